#!/usr/bin/env python3
"""Window variants for profiles/ubench/kloop_ubench.hip (what limits the MFMA issue rate of the wide MLP machinery?).
Each variant is a loop body of TWO windows (accumulator sets 0 / 1 alternate, so nothing is copied):
  bare2 / bare3       K loop only, 2 / 3 accumulator chains (3: no back-to-back dependent MFMAs)
  nolds2 / nolds3     the same without any LDS read (fragments never change)
  pf3_2               bare2 with weight fragments 3 K steps ahead instead of 2
  epi2 / epi3         K loop + the forward epilogue of the other accumulator set in the MFMA shadows (4 VALU per slot)
  epi2d               epi2 with the sigma' part (rcp, unorm16, 2 stores)
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nrhints_amd", "csrc"))
import gen_mlp32 as g

def body(chains, use_ds, pf, epi, want_d=False, nv=4, dma=False):
    out = ["{"]
    for w in range(2):
        cur, prev = w, 1 - w
        out.append("  {")
        out.append("    UB_TOP();")
        out.append("    const uint32_t wa = UB_WADDR();")
        win = g.Window(16, f"h{cur}", f"c{cur}", hh_zero=True, pf=pf, cd=(f"d{cur}" if chains == 3 else None), use_ds=use_ds)
        slots = None
        tail = []
        if epi:
            cp = f"c{prev}"
            ops = []
            if chains == 3:
                for r in range(16):
                    ops.append(g.Op(f"float s{r} = c{prev}[{r}] + d{prev}[{r}];", defs=(f"s{r}",)))
                e = g.epi_fwd(w, f"h{prev}", "S", want_d)
                for o in e:   # t = fma(s_r, LU, hh)
                    o.code = o.code.replace("S[", "s").replace("]", "", 1) if "S[" in o.code else o.code
                # dependencies: t_r uses s_r
                for o in e:
                    if o.code.startswith("float t"):
                        r = o.defs[0][1:]
                        o.uses = (f"s{r}",)
                ops += e
            else:
                ops = g.epi_fwd(w, f"h{prev}", cp, want_d)
            out.append(f'    asm volatile("" : "+v"(h{prev}), "+v"(c{prev}));')
            budget = (lambda k: max(1, nv - 2) if (dma and k in g.DMA_SLOTS16) else nv)
            slots, tail = g.schedule(ops, 48, budget)
        win.emit(out, slots, "    ", dma=(g.DMA_SLOTS16 if dma else None))
        if tail:
            g.emit_ops(out, tail, "    ")
        out.append("  }")
    out.append("}")
    return "\n".join(out) + "\n"

def main():
    d = os.path.join(ROOT, "profiles", "ubench", "gen")
    os.makedirs(d, exist_ok=True)
    files = {
        "bare2.inc": body(2, True, 2, False), "bare3.inc": body(3, True, 2, False),
        "nolds2.inc": body(2, False, 2, False), "nolds3.inc": body(3, False, 2, False),
        "pf3_2.inc": body(2, True, 3, False),
        "epi2.inc": body(2, True, 2, True), "epi3.inc": body(3, True, 2, True),
        "epi2d.inc": body(2, True, 2, True, want_d=True, nv=5),
        "epi2_nv6.inc": body(2, True, 2, True, nv=6),
        "bare2_dma.inc": body(2, True, 2, False, dma=True),
        "epi2_dma.inc": body(2, True, 2, True, dma=True),
        "epi2d_dma.inc": body(2, True, 2, True, want_d=True, nv=5, dma=True),
    }
    for k, v in files.items():
        open(os.path.join(d, k), "w").write(v)
    print("wrote", len(files), "variants")

if __name__ == "__main__":
    main()
