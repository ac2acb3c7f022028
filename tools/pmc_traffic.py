#!/usr/bin/env python3
"""HBM traffic per launch from rocprofv3 PMC passes, with the gfx950 FETCH_SIZE / WRITE_SIZE calibration applied.

Passes (each its own run, `--kernel-trace --pmc <one counter>` only, as MI355X_MICROARCH.md prescribes):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d D/calib_fetch -- tools/micro/fetch_calib
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d D/calib_write -- tools/micro/fetch_calib
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d D/conv_fetch  -- python3 tools/conv_only.py conv 4
    ... WRITE_SIZE -> D/conv_write;  the same for `costvol 4` -> D/cv_fetch, D/cv_write, `wino 4` -> D/wino_fetch, D/wino_write
    and `wino_wgrad 4` -> D/ww_fetch, D/ww_write
usage: pmc_traffic.py D > profiles/r02_pmc_traffic.json

Calibration: tools/micro/fetch_calib streams a known byte count once from a 1 GiB buffer with each access shape; factor =
known bytes / (counter * 1024).  The conv kernels stage their halo tiles with 4-byte-per-lane buffer loads in rows of 34
floats, so their FETCH_SIZE is scaled by the `read_b32_rows` factor computed against the 128-B lines those rows touch (the
memory system moves whole lines); 16-byte-per-lane streams (cost volume, GroupNorm) use the `read_b128_*` factor; the Winograd
convolution's 8-byte pair loads are contiguous across the lanes and take the coalesced `read_b32_buffer` factor (all
coalesced widths calibrate to the same 2.0)."""
import collections
import csv
import glob
import json
import sys

GIB = 1 << 30
ROWS = ((1 << 28) // 240)
KNOWN = {"read_b32_buffer": GIB, "read_b32_rows_payload": ROWS * 34 * 4, "read_b32_rows_lines": ROWS * 256,
         "read_b128_buffer": GIB, "read_b128_global": GIB, "write_b128": GIB}


def counters(d):
    files = glob.glob(f"{d}/*/*_counter_collection.csv")
    per = collections.defaultdict(list)
    if files:
        for r in csv.DictReader(open(files[0])):
            per[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return per


def mean_kib(per, kernel_sub, counter, skip_first=0):
    vals = [v for (k, c), vs in per.items() if kernel_sub in k and c == counter for v in vs[skip_first:]]
    return sum(vals) / len(vals) if vals else None


def main():
    root = sys.argv[1]
    cf, cw = counters(f"{root}/calib_fetch"), counters(f"{root}/calib_write")
    out = {"_how": __doc__, "calibration": {}}
    fac = {}
    for name in ("read_b32_buffer", "read_b32_rows", "read_b128_buffer", "read_b128_global"):
        kib = mean_kib(cf, name, "FETCH_SIZE")
        if kib is None:
            continue
        if name == "read_b32_rows":
            out["calibration"][name] = {"FETCH_SIZE_KiB": kib, "payload_bytes": KNOWN["read_b32_rows_payload"],
                                        "line_bytes": KNOWN["read_b32_rows_lines"],
                                        "factor_vs_lines": KNOWN["read_b32_rows_lines"] / (kib * 1024),
                                        "factor_vs_payload": KNOWN["read_b32_rows_payload"] / (kib * 1024)}
            fac[name] = KNOWN["read_b32_rows_lines"] / (kib * 1024)
        else:
            out["calibration"][name] = {"FETCH_SIZE_KiB": kib, "known_bytes": KNOWN[name], "factor": KNOWN[name] / (kib * 1024)}
            fac[name] = KNOWN[name] / (kib * 1024)
    kib = mean_kib(cw, "write_b128", "WRITE_SIZE")
    if kib:
        out["calibration"]["write_b128"] = {"WRITE_SIZE_KiB": kib, "known_bytes": GIB, "factor": GIB / (kib * 1024)}
        fac["write_b128"] = GIB / (kib * 1024)
    wf = fac.get("write_b128", 1.0)
    for key, sub, dirs, ffac, alg in (
            ("conv3d_k3_mfma_32to32_B4", "conv3d_k3_mfma", ("conv_fetch", "conv_write"), fac.get("read_b32_rows", 1.0),
             2 * 4 * 32 * 48 * 144 * 240 * 4),
            # the Winograd kernel fetches its patches as 8-byte pairs contiguous across the lanes: the coalesced factor
            ("conv_wino_mfma_32to32_B4", "conv_wino_mfma", ("wino_fetch", "wino_write"), fac.get("read_b32_buffer", 2.0),
             2 * 4 * 32 * 48 * 144 * 240 * 4),
            ("conv3d_wgrad_wino_32to32_B4", "conv3d_wgrad_mfma", ("ww_fetch", "ww_write"), fac.get("read_b32_rows", 1.0),
             2 * 4 * 32 * 48 * 144 * 240 * 4),
            # ECM weights backward (round 3): hr read twice + saved planes + their gradient (4-byte-per-lane global loads, rows of
            # 16 consecutive floats), ghr + gA9 written
            ("ecm_weights_bwd_kernel_p_B4", "ecm_weights_bwd_kernel_p", ("ew_fetch", "ew_write"), fac.get("read_b32_buffer", 2.0),
             4 * (3 * 32 + 18) * 576 * 960 * 4),
            # the classifier forward (round 4: with GroupNorm + ReLU on load): 34-float halo rows of a blocked 3-D tile, 4-byte
            # lanes -> the row factor (bytes of the 128-B lines those rows touch); algorithmic = x once + y
            ("conv3d_c1_fwd_v_B4", "conv3d_c1_fwd_v", ("c1_fetch", "c1_write"), fac.get("read_b32_rows", 1.0),
             4 * (32 + 1) * 48 * 144 * 240 * 4),
            ("costvol_fwd_v4_B4", "costvol_fwd", ("cv_fetch", "cv_write"), fac.get("read_b128_global", 2.0),
             4 * (2 * 32 * 48 * 144 * 240 + 2 * 32 * 144 * 240) * 4)):
        f = mean_kib(counters(f"{root}/{dirs[0]}"), sub, "FETCH_SIZE", 1)
        w = mean_kib(counters(f"{root}/{dirs[1]}"), sub, "WRITE_SIZE", 1)
        if f is None or w is None:
            continue
        rd, wr = f * 1024 * ffac, w * 1024 * wf
        out[key] = {"FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB_raw": w, "read_bytes_calibrated": rd, "write_bytes_calibrated": wr,
                    "hbm_bytes_per_launch": rd + wr, "algorithmic_bytes": alg, "ratio_to_algorithmic": (rd + wr) / alg}
    out["note"] = ("rocprofv3 PMC (FETCH_SIZE / WRITE_SIZE in separate passes), B=4, calibrated with tools/micro/fetch_calib "
                   "(tools/pmc_traffic.py)")
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
