#!/usr/bin/env python3
"""Summarise `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d <dir> -- python3 tools/conv_only.py <which> <B>`
runs into matrix-pipe utilisation per kernel.  usage: pmc_mfma_summary.py tag=dir [tag=dir ...] > profiles/<name>.json

SQ_VALU_MFMA_BUSY_CYCLES sums, over all SIMDs, the cycles a matrix pipe is busy (64 per v_mfma_f32_32x32x2_f32);
GRBM_GUI_ACTIVE is summed over the 8 XCDs.  utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs).

NOT a clock measurement: a launch's duration in a --pmc pass includes the counters' start / stop around it (the 1.58 ms
convolution reads 1.92 ms there), so GUI_ACTIVE / 8 / that duration UNDER-states the clock (1.9 GHz for a kernel that runs at
2.3-2.4: its active cycles / 8 divided by its duration WITHOUT counters give 2.31 GHz, and tools/micro/mfma_power, whose
launches last 0.35 s, reads 2.38 GHz by the same division with or without side work -- profiles/r04_mfma_power.txt).  The
figure is kept as `gui_active_per_pmc_duration_GHz` with that caveat; `active_ms_at_2p38GHz` = active cycles at the clock
the long launches show, which agrees with the un-instrumented launch time."""
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
            "gui_active_per_pmc_duration_GHz": round(cyc / (ms * 1e6), 3),
            "active_ms_at_2p38GHz": round(cyc / 2.38e6, 4),
            "mfma_util": round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 4),
            "mfma_instructions_32x32x2": round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / 64),
        }
json.dump(out, sys.stdout, indent=1)
print()
