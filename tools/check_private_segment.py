#!/usr/bin/env python3
"""Private-segment (scratch memory) audit of every kernel in libecm_hip.so.

A kernel with a non-zero `.private_segment_fixed_size` makes the runtime provision scratch memory at DISPATCH time: bytes
per lane x 64 lanes x every wave slot of the device.  Round 2's `ecm_weights_bwd_kernel<0>` (65 spilled VGPRs) was the only
such kernel of the library and the only one that ever aborted the process with no message (`r2_t8.log`); DESIGN.md section 4
works the numbers.  This tool reads the AMDGPU metadata notes of the gfx950 code objects embedded in the library and

  * prints name / private segment / spilled SGPRs+VGPRs / VGPR+AGPR count of every kernel with `-v`,
  * FAILS (exit 1) if ANY kernel has a non-zero private segment or a non-zero VGPR spill count -- so a spilling kernel
    cannot enter the library unnoticed (run by __graft_entry__.build() and tests/test_abi.py).

usage: check_private_segment.py [-v] [libecm_hip.so]
"""
import os
import re
import subprocess
import sys

LLVM = "/opt/rocm/lib/llvm/bin"
HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_SO = os.path.join(HERE, "..", "explicit-context-mapping-for-stereo-matching_amd", "csrc", "libecm_hip.so")
FIELDS = (".private_segment_fixed_size", ".sgpr_spill_count", ".vgpr_spill_count", ".vgpr_count", ".agpr_count", ".sgpr_count",
          ".group_segment_fixed_size")


def kernels_of(so):
    """-> [dict(name=..., private_segment_fixed_size=..., ...)] over every gfx950 code object embedded in `so`"""
    tmp = subprocess.run(["mktemp", "-d"], capture_output=True, text=True, check=True).stdout.strip()
    out = []
    try:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat], check=True)
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        for n, st in enumerate(starts):
            part = os.path.join(tmp, f"bundle{n}.bin")
            with open(part, "wb") as f:
                f.write(blob[st:starts[n + 1] if n + 1 < len(starts) else len(blob)])
            co = os.path.join(tmp, f"gfx950_{n}.co")
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            cur = None
            for line in notes.splitlines():
                t = line.strip()
                if t.startswith("- .agpr_count") or (t.startswith("- .") and cur is not None and ".name" not in cur):
                    pass
                if t.startswith("- ."):                      # first key of a new list element (kernel entries start this way)
                    if cur and "name" in cur:
                        out.append(cur)
                    cur = {}
                    t = t[2:]
                if cur is None:
                    continue
                m = re.match(r"(\.[a-z_]+):\s+(.*)$", t)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip().strip("'\"")
                if k == ".name" and "name" not in cur and not v.isdigit():
                    cur["name"] = v
                elif k in FIELDS:
                    try:
                        cur[k[1:]] = int(v)
                    except ValueError:
                        pass
            if cur and "name" in cur:
                out.append(cur)
    finally:
        subprocess.run(["rm", "-rf", tmp])
    # argument entries also carry `.name`; kernels are the entries that have a private segment field
    return [k for k in out if "private_segment_fixed_size" in k]


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    except OSError:
        return names
    return r.stdout.splitlines() if r.returncode == 0 and len(r.stdout.splitlines()) == len(names) else names


def main():
    args = [a for a in sys.argv[1:] if a != "-v"]
    verbose = "-v" in sys.argv[1:]
    so = args[0] if args else DEFAULT_SO
    ks = kernels_of(so)
    if not ks:
        print("no gfx950 kernels found in", so)
        return 1
    names = demangle([k["name"] for k in ks])
    bad = []
    for k, nm in zip(ks, names):
        ps, vs = k.get("private_segment_fixed_size", 0), k.get("vgpr_spill_count", 0)
        if verbose or ps or vs:
            print(f"{ps:6d} B private  {vs:4d} vgpr spills {k.get('sgpr_spill_count', 0):4d} sgpr spills  "
                  f"{k.get('vgpr_count', 0):3d}+{k.get('agpr_count', 0):3d} regs  {k.get('group_segment_fixed_size', 0):6d} B lds  {nm[:150]}")
        if ps or vs:
            bad.append(nm)
    print(f"{len(ks)} kernels audited, {len(bad)} with a private segment / VGPR spills")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
