#!/usr/bin/env python3
"""Summarise `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d <dir> -- python3 tools/conv_only.py <which> <B>`
runs into matrix-pipe utilisation per kernel.  usage: pmc_mfma_summary.py tag=dir [tag=dir ...] > profiles/<name>.json

SQ_VALU_MFMA_BUSY_CYCLES sums, over all SIMDs, the cycles a matrix pipe is busy (64 per v_mfma_f32_32x32x2_f32);
GRBM_GUI_ACTIVE is summed over the 8 XCDs.  utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs);
clock = GUI_ACTIVE / 8 / duration."""
import collections
import csv
import glob
import json
import sys

out = {"_how": __doc__}
for arg in sys.argv[1:]:
    tag, d = arg.rsplit("=", 1)
    cc = list(csv.DictReader(open(glob.glob(f"{d}/*/*_counter_collection.csv")[0])))
    tr = list(csv.DictReader(open(glob.glob(f"{d}/*/*_kernel_trace.csv")[0])))
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in cc:
        per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for r in tr:
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for k, c in per.items():
        if "mfma" not in k:
            continue
        m = {n: sum(v) / len(v) for n, v in c.items()}
        ms = sum(dur[k]) / len(dur[k])
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        out[f"{tag}: {k.replace('void (anonymous namespace)::', '').split('(float')[0]}"] = {
            "launches": len(dur[k]), "avg_ms_under_pmc": round(ms, 4), "counters_per_launch": m,
            "shader_clock_GHz": round(cyc / (ms * 1e6), 3),
            "mfma_util": round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 4),
            "mfma_instructions_32x32x2": round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / 64),
        }
json.dump(out, sys.stdout, indent=1)
print()
