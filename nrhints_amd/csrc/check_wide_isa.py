#!/usr/bin/env python3
"""Static checks on the wide kernels' ISA (hipcc -save-temps .s) for the things the compiler is trusted NOT to do:
  1. no v_accvgpr_read / v_accvgpr_mov and no AGPR spills (AGPRs belong to the generated schedule);
  2. no scratch use, no packed-f32 VALU;
  3. a register written by an asm global_load (q words) is not read or written by anything before the next s_waitcnt vmcnt
     that covers it - here: before the next `s_waitcnt vmcnt(` at all (conservative);
  4. a half-register write (v_fma_mixlo_f16 / v_fma_mixhi_f16) is not consumed by the very next instruction;
  5. no scalar-memory instruction (s_load / s_buffer_load / s_memtime) behind a kernel's first MFMA: SMEM results return out of order
     with LDS reads and count in the same lgkmcnt, so one in flight inside a window would let a computed `lgkmcnt(N)` pass with a
     weight fragment still on its way (the timing builds of profiles/ubench, which stamp with s_memtime, are exempt - and showed it).
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
    nload, npartial, partial, seen_mfma = 0, 0, None, False
    for ln, line in enumerate(open(path), 1):
        m = re.match(r"^(_ZN\d+nrh32t?(?:12sdf32_kernelILi\dE|14color32_kernelE)\w+):", line)       # nrh32: three-term builds, nrh32t: one-term
        if m: kern, inflight, partial, seen_mfma = m.group(1), {}, None, False; continue
        if kern is None: continue
        if line.startswith(".Lfunc_end"): kern = None; continue
        t = line.strip().replace(",", " ").split()
        if not t or t[0][0] in ";.": continue
        op = t[0]
        if op.startswith("v_mfma"): seen_mfma = True
        if seen_mfma and op.startswith(("s_load_", "s_buffer_load", "s_memtime", "s_memrealtime")):
            bad.append((ln, "scalar memory instruction behind the first MFMA (shares lgkmcnt with the LDS reads, returns out of order)", line.strip()))
        if op.startswith(("v_accvgpr_read", "v_accvgpr_mov")): bad.append((ln, "AGPR read/mov", line.strip()))
        if op.startswith("scratch_") or (op.startswith("v_pk_") and "f32" in op): bad.append((ln, "scratch / packed f32", line.strip()))
        if op.startswith("s_waitcnt") and "vmcnt" in line: inflight, partial = {}, None; continue
        used = set()
        for tok in t[1:]: used |= regs_of(tok)
        # 4. half-register writes (v_fma_mixlo_f16 / v_fma_mixhi_f16 of the generated hi / lo split, inside asm statements hipcc's
        #    hazard recognizer does not see): the next instruction must not touch that register (gfx940+ dst_sel forwarding: 1 wait state)
        if partial is not None and partial in used: bad.append((ln, f"v{partial}: half-register write consumed by the next instruction", line.strip()))
        partial = None
        if op in ("v_fma_mixlo_f16", "v_fma_mixhi_f16"):
            npartial += 1
            partial = min(regs_of(t[1]))
        hit = used & set(inflight)
        if hit: bad.append((ln, f"touches in-flight load result v{sorted(hit)[0]} (loaded at line {inflight[sorted(hit)[0]]})", line.strip()))
        if op.startswith("global_load_dwordx4"):
            nload += 1
            for r in regs_of(t[1]): inflight[r] = ln
    print(f"{path}: {nload} asm/global dwordx4 loads, {npartial} half-register writes checked, {len(bad)} problem(s)")
    for b in bad[:20]: print("  line %d: %s: %s" % b)
    return 1 if bad else 0

def check_inflight_loads(path, kernel_regex):
    """For kernels whose asm loads stay in flight ACROSS other waits (csrc/nrh_dw.hip: two K steps of loads outstanding): model
    vmcnt as the in-order counter it is.  Every global_load appends its destination registers to the queue; `s_waitcnt vmcnt(N)`
    retires all but the youngest N; any instruction in between that reads or writes a register of a queued load is reported.
    The scan is linear in the text (block layout order): the kernel's loop is laid out in execution order, and the queue is
    emptied at every vmcnt(0)."""
    kern, bad, queue, nload = None, [], [], 0
    for ln, line in enumerate(open(path), 1):
        m = re.match(r"^(\w*" + kernel_regex + r"\w*):", line)
        if m: kern, queue = m.group(1), []; continue
        if kern is None: continue
        if line.startswith(".Lfunc_end"): kern = None; continue
        t = line.strip().replace(",", " ").split()
        if not t or t[0][0] in ";.": continue
        op = t[0]
        if op.startswith("s_waitcnt"):
            m = re.search(r"vmcnt\((\d+)\)", line)
            if m:
                n = int(m.group(1))
                queue = queue[len(queue) - n:] if n else []
            continue
        used = set()
        for tok in t[1:]: used |= regs_of(tok)
        for regs, at in queue:
            hit = used & regs
            if hit:
                bad.append((ln, f"touches v{sorted(hit)[0]}, still in flight from the load at line {at}", line.strip()))
                break
        if op.startswith("global_load"):
            nload += 1
            queue.append((regs_of(t[1]), ln))
        elif op.startswith(("global_store", "global_atomic")):
            queue.append((set(), ln))          # stores count in vmcnt too
    print(f"{path}: {kernel_regex}: {nload} global loads checked against the in-order vmcnt model, {len(bad)} problem(s)")
    for b in bad[:20]: print("  line %d: %s: %s" % b)
    return 1 if bad else 0


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--inflight":
        sys.exit(check_inflight_loads(sys.argv[1], sys.argv[3]))
    sys.exit(main())
