#!/usr/bin/env python3
"""Static checks on the wide kernels' ISA (hipcc -save-temps .s) for the things the compiler is trusted NOT to do:
  1. no v_accvgpr_read / v_accvgpr_mov and no AGPR spills (AGPRs belong to the generated schedule);
  2. no scratch use, no packed-f32 VALU;
  3. a register written by an asm global_load (q words) is not read or written by anything before the next s_waitcnt vmcnt
     that covers it - here: before the next `s_waitcnt vmcnt(` at all (conservative).
    python nrhints_amd/csrc/check_wide_isa.py file.s   (the Makefile runs it on every build of nrh_wide.o)
"""
import re, sys

def regs_of(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()

def main():
    path = sys.argv[1]
    kern, bad, inflight = None, [], {}
    nload = 0
    for ln, line in enumerate(open(path), 1):
        m = re.match(r"^(_ZN5nrh32(?:12sdf32_kernelILi\dE|14color32_kernelE)\w+):", line)
        if m: kern, inflight = m.group(1), {}; continue
        if kern is None: continue
        if line.startswith(".Lfunc_end"): kern = None; continue
        t = line.strip().replace(",", " ").split()
        if not t or t[0][0] in ";.": continue
        op = t[0]
        if op.startswith(("v_accvgpr_read", "v_accvgpr_mov")): bad.append((ln, "AGPR read/mov", line.strip()))
        if op.startswith("scratch_") or (op.startswith("v_pk_") and "f32" in op): bad.append((ln, "scratch / packed f32", line.strip()))
        if op.startswith("s_waitcnt") and "vmcnt" in line: inflight = {}; continue
        used = set()
        for tok in t[1:]: used |= regs_of(tok)
        hit = used & set(inflight)
        if hit: bad.append((ln, f"touches in-flight load result v{sorted(hit)[0]} (loaded at line {inflight[sorted(hit)[0]]})", line.strip()))
        if op.startswith("global_load_dwordx4"):
            nload += 1
            for r in regs_of(t[1]): inflight[r] = ln
    print(f"{path}: {nload} asm/global dwordx4 loads checked, {len(bad)} problem(s)")
    for b in bad[:20]: print("  line %d: %s: %s" % b)
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
