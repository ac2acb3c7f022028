#!/usr/bin/env python3
"""Static check of the Winograd kernels' hand-issued patch loads (csrc/conv_wino.hip).

The patch loads are inline asm (`buffer_load_dwordx2`), waited for by hand-counted `s_waitcnt vmcnt(N)`: the compiler treats
their destination registers as valid from the moment of issue.  This walks the disassembly of every `conv_wino_mfma`
instantiation and verifies, by a forward data-flow analysis over the control-flow graph, that no instruction reads or
overwrites a destination register of such a load before a wait that proves it landed:

  * a load L is complete after `s_waitcnt vmcnt(N)` iff at most N VMEM operations were issued after L before that wait;
  * until then the registers of L may appear in no other instruction.

usage: check_wino_isa.py [libecm_hip.so]   (exit code 0 = clean; prints the offending lines otherwise)
"""
import os
import re
import subprocess
import sys

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_SO = os.path.join(HERE, "..", "explicit-context-mapping-for-stereo-matching_amd", "csrc", "libecm_hip.so")

VMEM = re.compile(r"^(buffer_|global_|flat_|scratch_)(load|store|atomic)")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def disassemble(so):
    """-> {kernel name: [instruction text, ...]} for the gfx950 code object embedded in the library"""
    tmp = subprocess.run(["mktemp", "-d"], capture_output=True, text=True, check=True).stdout.strip()
    txt = ""
    try:
        # .hip_fatbin holds one offload bundle per linked object: split at the bundle magic, unbundle each, keep gfx950
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat], check=True)
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        for n, st in enumerate(starts):
            part = os.path.join(tmp, f"bundle{n}.bin")
            with open(part, "wb") as f:
                f.write(blob[st:starts[n + 1] if n + 1 < len(starts) else len(blob)])
            co = os.path.join(tmp, f"gfx950_{n}.co")
            r = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            d = subprocess.run([OBJDUMP, "-d", co], capture_output=True, text=True)
            if "conv_wino_mfma" in d.stdout:
                txt += d.stdout
    finally:
        subprocess.run(["rm", "-rf", tmp])
    kernels, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        if cur is None or "//" not in line:
            continue
        ins, _, rest = line.partition("//")
        ins = ins.strip()
        m = re.match(r"\s*([0-9A-Fa-f]+):", rest)
        if ins and m:
            kernels[cur].append((int(m.group(1), 16), ins))
    return kernels


def branch_target(addr, text):
    parts = text.split()
    if not (parts[0].startswith("s_cbranch") or parts[0] == "s_branch"):
        return None
    imm = int(parts[1]) & 0xFFFF
    imm -= 0x10000 if imm & 0x8000 else 0
    return addr + 4 + 4 * imm


CAP = 64                           # vmcnt saturates at 63: an age of 64 means "done whatever the wait"


def transfer(state, text, problems, where):
    """state: {register: number of VMEM operations issued after the asm load that writes it} for loads possibly in flight"""
    op = text.split()[0]
    if op == "s_waitcnt":
        m = re.search(r"vmcnt\((\d+)\)", text)
        if m:                      # in-order return: at most n outstanding -> a load with >= n younger operations is done
            n = int(m.group(1))
            state = {r: a for r, a in state.items() if a < n}
        return state
    if op == "s_endpgm":
        return {}
    touched = regs_of(text)
    if text.startswith("buffer_load_dwordx2"):
        dst = regs_of(text.split(",")[0])
        bad = (touched - dst) & set(state)
        if bad and problems is not None:
            problems.append(f"{where} {text}  uses in-flight v{sorted(bad)}")
        state = {r: min(a + 1, CAP) for r, a in state.items() if a + 1 < CAP}
        for r in dst:
            state[r] = 0
        return state
    bad = touched & set(state)
    if bad and problems is not None:
        problems.append(f"{where} {text}  touches in-flight v{sorted(bad)}")
    if VMEM.match(op):
        state = {r: a + 1 for r, a in state.items() if a + 1 < CAP}
    return state


def check_kernel(name, code):
    """Forward data-flow over the control-flow graph; join = per register the smallest age (fewest younger operations)."""
    if not any(t.startswith("buffer_load_dwordx2") for _, t in code):
        return [f"{name}: no buffer_load_dwordx2 found (check is stale)"]
    index_of = {addr: i for i, (addr, _) in enumerate(code)}
    leaders = {0}
    for i, (addr, t) in enumerate(code):
        tgt = branch_target(addr, t)
        if tgt is not None:
            if tgt in index_of:
                leaders.add(index_of[tgt])
            if i + 1 < len(code):
                leaders.add(i + 1)
    starts = sorted(leaders)
    blocks = {st: (st, starts[n + 1] if n + 1 < len(starts) else len(code)) for n, st in enumerate(starts)}
    succ = {}
    for st, (lo, hi) in blocks.items():
        addr, t = code[hi - 1]
        out = []
        tgt = branch_target(addr, t)
        op = t.split()[0]
        if tgt is not None and tgt in index_of:
            out.append(index_of[tgt])
        if op not in ("s_branch", "s_endpgm") and hi < len(code):
            out.append(hi)
        succ[st] = out
    entry = {st: None for st in blocks}          # None = not reached yet
    entry[0] = {}
    work = [0]
    while work:
        st = work.pop()
        state = dict(entry[st])
        lo, hi = blocks[st]
        for i in range(lo, hi):
            state = transfer(state, code[i][1], None, "")
        for nx in succ[st]:
            old = entry[nx]
            if old is None:
                merged = dict(state)
            else:
                merged = dict(old)
                for r, a in state.items():
                    merged[r] = min(a, merged.get(r, CAP))
            if merged != old:
                entry[nx] = merged
                work.append(nx)
    problems = []
    for st, (lo, hi) in blocks.items():
        if entry[st] is None:
            continue
        state = dict(entry[st])
        for i in range(lo, hi):
            state = transfer(state, code[i][1], problems, f"{name}: [{i}]")
    return problems


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else DEFAULT_SO
    kernels = disassemble(so)
    names = [k for k in kernels if "conv_wino_mfma" in k]
    if not names:
        print("no conv_wino_mfma kernels in", so)
        return 2
    problems = []
    for k in names:
        problems += check_kernel(k, kernels[k])
    for p in problems[:40]:
        print(p)
    print(f"{len(names)} kernels checked, {len(problems)} problems")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
