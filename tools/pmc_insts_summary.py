#!/usr/bin/env python3
"""Instruction mix per launch and per wave of the Winograd kernels from a `rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU
SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES` pass (tools/prof_round.sh).
usage: pmc_insts_summary.py tag=dir [tag=dir ...] > profiles/<name>.json

SQ_INSTS_VALU counts the MFMA instructions too; the number of MFMAs per wave is known from the kernel's shape
(chunks x MFMAs per chunk), so the non-matrix VALU instructions per MFMA follow -- the quantity DESIGN.md section 3 argues
bounds these kernels (a wave's VALU and MFMA instructions do not overlap)."""
import collections
import csv
import glob
import json
import sys

MFMA_PER_WAVE = {"conv_wino_mfma<3, 2, 1, 2>": 16 * 24, "conv_wino_mfma<1, 1, 2, 4>": None}    # 2-D: Ci / 4 chunks x 16

out = {"_how": __doc__}
for arg in sys.argv[1:]:
    tag, d = arg.rsplit("=", 1)
    f = glob.glob(f"{d}/*/*_counter_collection.csv")
    if not f:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "conv_wino_mfma" in r["Kernel_Name"]:
            per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in per.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        waves = m.get("SQ_WAVES", 0.0)
        row = {"counters_per_launch": m}
        if waves:
            row["per_wave"] = {n[8:].lower(): round(v / waves, 1) for n, v in m.items() if n.startswith("SQ_INSTS_")}
        out[f"{tag}: {k.replace('void (anonymous namespace)::', '').split('(float')[0]}"] = row
json.dump(out, sys.stdout, indent=1)
print()
