#!/usr/bin/env python3
"""Static look at a gfx950 .s file (hipcc -save-temps): per kernel, resource metadata, instruction-class census, packed-f32
count, and for every basic block with MFMAs the issue-slot mix per MFMA (VALU / transcendental / DS / VMEM / SALU / waits).

    python profiles/tools/isa_stats.py file.s [kernel-substring] [--blocks]
"""
import re, sys, collections

def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_accvgpr"): return "acc"
    if op in ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32") or \
       op.startswith(("v_exp_f32", "v_log_f32", "v_rcp_f32")): return "trans"
    if op.startswith("v_pk_") and op.endswith("f32"): return "pkf32"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "ds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    return "other"

def main():
    path = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
    show_blocks = "--blocks" in sys.argv
    kernels, cur, name = {}, None, None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1); cur = kernels.setdefault(name, {"blocks": [["entry", []]], "meta": {}}); continue
        if cur is None: continue
        s = line.strip()
        if s.startswith(".end_amdhsa_kernel") or s.startswith(".section"): cur = None; continue
        m = re.match(r"^(\.LBB\w+):", s)
        if m: cur["blocks"].append([m.group(1), []]); continue
        if not s or s.startswith((";", ".", "//")):
            m = re.match(r"; (\w[\w ]*): (\d+)", s)
            if m: cur["meta"][m.group(1)] = int(m.group(2))
            continue
        op = s.split()[0]
        cur["blocks"][-1][1].append((op, s))
    for name, k in kernels.items():
        if pat not in name: continue
        tot = collections.Counter()
        for _, ins in k["blocks"]:
            for op, _ in ins: tot[classify(op)] += 1
        meta = k["meta"]
        keys = ["NumVgprs", "NumAgprs", "TotalNumVgprs", "ScratchSize", "Occupancy", "NumSgprs"]
        print(f"== {name}\n   " + " ".join(f"{q}={meta.get(q)}" for q in keys if q in meta))
        print("   static:", dict(tot))
        for label, ins in k["blocks"]:
            c = collections.Counter(classify(op) for op, _ in ins)
            if c["mfma"] < 8: continue
            n = c["mfma"]
            per = {q: round(c[q] / n, 2) for q in ("valu", "trans", "pkf32", "acc", "ds", "vmem", "salu", "wait", "nop")}
            print(f"   block {label}: {len(ins)} instr, {n} mfma; per mfma: {per}")
            if show_blocks:
                gaps, g = [], collections.Counter()
                for op, _ in ins:
                    cl = classify(op)
                    if cl == "mfma":
                        gaps.append(g); g = collections.Counter()
                    else: g[cl] += 1
                line = " ".join("".join(f"{q[0]}{v}" for q, v in sorted(x.items())) or "-" for x in gaps)
                print("      gaps:", line)

if __name__ == "__main__":
    main()
