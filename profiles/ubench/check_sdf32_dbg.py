#!/usr/bin/env python3
"""Compare gpurun_out/sdf32_dbg.bin (intermediate state of workgroup 0's first pass, written by the -DNRH32_DEBUG build of
sdf32_bench) with the numpy emulation of the same tile (tests/mfma32_emulator.py), stage by stage."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import mfma32_emulator as emu
from nrhints_amd import packing32 as pk32

def main():
    case = os.path.join(ROOT, "profiles/ubench/data/sdf32_case.bin")
    raw = open(case, "rb").read()
    nrays, nper, ncheck, nstream, ntab = np.frombuffer(raw[:40], dtype=np.int64)
    off = 40
    streams = np.frombuffer(raw[off:off + nstream * 2], dtype=np.float16); off += nstream * 2
    tab = np.frombuffer(raw[off:off + ntab * 4], dtype=np.float32).reshape(-1, 256); off += ntab * 4
    ro = np.frombuffer(raw[off:off + nrays * 12], dtype=np.float32).reshape(-1, 3); off += nrays * 12
    rd = np.frombuffer(raw[off:off + nrays * 12], dtype=np.float32).reshape(-1, 3); off += nrays * 12
    t = np.frombuffer(raw[off:off + nrays * nper * 4], dtype=np.float32).reshape(nrays, nper)
    dbg = np.fromfile(os.path.join(ROOT, "gpurun_out/sdf32_dbg.bin"), dtype=np.uint32).reshape(7, 32768)
    s0 = streams[:pk32.stream_bytes(0) // 2]
    def halves(w):  # uint32 [..] -> two float64 from packed fp16
        h = np.ascontiguousarray(w).view(np.float16).astype(np.float64)
        return h[..., 0::2], h[..., 1::2]
    for wave in range(4):
        pts = (ro[0] + rd[0] * t[0, wave * 32: wave * 32 + 32, None]).astype(np.float32).astype(np.float64)   # ray 0, samples 32w..
        tr = {}
        emu.sdf32_tile(s0, tab, pts, 0, trace=tr)
        # section 1: embedding B operands, words [hi s0 p0..3, s1, s2, lo ...]
        w = dbg[1].reshape(4, 128, 64)[wave][:24]          # [24, 64]
        hi = w[:12].reshape(3, 4, 64); lo = w[12:].reshape(3, 4, 64)
        err = 0.0
        for s in range(3):
            for p in range(4):
                h0, h1 = halves(hi[s, p]); l0, l1 = halves(lo[s, p])
                gv0, gv1 = h0 + l0 / 2048, h1 + l1 / 2048
                ev0 = tr['eb'][0][s][:, 2 * p] + tr['eb'][1][s][:, 2 * p] / 2048
                ev1 = tr['eb'][0][s][:, 2 * p + 1] + tr['eb'][1][s][:, 2 * p + 1] / 2048
                err = max(err, np.abs(gv0 - ev0).max(), np.abs(gv1 - ev1).max())
        print(f"wave {wave}: S1 embedding max err {err:.3e}")
        # sections 3..6: `in` after L0, L1, L2, L7: a[i] for i in 0..127: hi words 0..63 (K step s = i//4), lo words 64..127
        for sec, layer in ((3, 0), (4, 1), (5, 2), (6, 7)):
            a = dbg[sec].reshape(4, 128, 64)[wave]
            u = tr['u'][layer]                       # [8 chunks][16, 64]
            err = 0.0; mx = 0.0
            for c in range(8):
                for i in range(8):
                    h0, h1 = halves(a[8 * c + i]); l0, l1 = halves(a[64 + 8 * c + i])
                    g0, g1 = h0 + l0 / 2048, h1 + l1 / 2048
                    err = max(err, np.abs(g0 - u[c][2 * i]).max(), np.abs(g1 - u[c][2 * i + 1]).max())
                    mx = max(mx, np.abs(u[c][2 * i]).max())
            print(f"wave {wave}: S{sec} in after L{layer}: max err {err:.3e} (|ref| max {mx:.3f})")

if __name__ == "__main__":
    main()
