#!/usr/bin/env python3
"""Stand-alone timing of the weight-gradient launch (csrc/nrh_dw.hip through nrhints_amd/dw.py) with the job table of one
1024-ray training step (P = 131 072 points: 8 SDF layers + 2 heads + 5 reflectance layers) on random data.
    python profiles/dw_bench.py [rays] [items,items,...]
Prints ms per call, the HBM read rate (operands read once: 5.7 GB) and the algorithmic TFLOP/s."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nrhints_amd import dw

def main():
    rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    items = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [256, 512, 384]
    P = rays * 128
    dev = torch.device("cuda")
    g = lambda *s: torch.randn(*s, device=dev) * 0.01
    h, t, zbar, abar = g(8, P, 256), g(8, P, 256), g(8, P, 256), g(8, P, 256)
    gebar, emb, sbar, fbar = g(P, 64), g(P, 64), g(P), g(P, 256)
    czbar, zbar4, save_h, feat, misc = g(4, P, 256), g(P, 3), g(4, P, 256), g(P, 256), g(P, 128)
    new = lambda *s: torch.empty(*s, device=dev)
    out = {}
    shapes = [(256, 39)] + [(256, 256)] * 2 + [(217, 256)] + [(256, 256)] * 4
    for l in range(8):
        out[f"dW{l}"], out[f"db{l}"] = new(*shapes[l]), new(shapes[l][0])
    out.update(ws=new(1, 256), bs=new(1), Wf=new(256, 256), bf=new(256), w0=new(256, 361), w1=new(256, 256), w2=new(256, 256), w3=new(256, 256),
               w4=new(3, 256), b0=new(256), b1=new(256), b2=new(256), b3=new(256), b4=new(3))
    jobs = dw.sdf_jobs(shapes, h, t, zbar, abar, gebar, emb, sbar, fbar, out) + dw.color_jobs(True, czbar, zbar4, save_h, feat, misc, out)
    nbytes = 4 * P * (7 * 4 * 256 + 2 * 256 + 2 * 64 + 2 * 256 + 2 * 256 + 2 + 2 * 256 + 256 + 128 + 3 * 2 * 256 + 256 + 3)
    flop = 2.0 * P * (7 * 2 * 65536 + 2 * 256 * 39 + 65536 + 2 * 256 + 65536 + 256 * 105 + 3 * 65536 + 256 * 3)
    for it in items:
        for _ in range(2):
            dw.run(jobs, P, total_items=it)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        n = 10
        for _ in range(n):
            dw.run(jobs, P, total_items=it)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / n
        lib = __import__("nrhints_amd")._lib.load()
        if hasattr(lib, "nrh_dw_debug_read"):
            import ctypes
            buf = (ctypes.c_ulonglong * 8)()
            lib.nrh_dw_debug_read(buf, 1)
            steps, fsteps = max(1, buf[5] & 0xffffffff), max(1, buf[5] >> 32)
            print("   general path, cycles per K step (wave 0 of every workgroup): issue %.0f | wait+barrier %.0f | compute %.0f | convert %.0f | barrier %.0f   (%d steps)"
                  % tuple([buf[k] / steps for k in range(5)] + [steps]))
            print("   fast path: block %.0f | wait+barrier %.0f   (%d steps)" % (buf[6] / fsteps, buf[7] / fsteps, fsteps))
        print(f"rays {rays} items {it}: {ms:.3f} ms per call  {nbytes / ms / 1e6:.0f} GB/s read  {flop / ms / 1e9:.1f} TFLOP/s algorithmic ({3 * flop / ms / 1e9:.0f} of MFMA issue)", flush=True)
    # reference point: the round-2 formulation of ONE full two-pair job in torch (fp32 rocBLAS, split into 64 batches)
    S = 64
    def big_k(a3, b3):
        return torch.bmm(a3.reshape(S, P // S, 256).transpose(1, 2), b3.reshape(S, P // S, 256)).sum(0)
    for _ in range(2):
        big_k(zbar[1], h[0]) + big_k(t[1], abar[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        big_k(zbar[1], h[0]) + big_k(t[1], abar[0])
    torch.cuda.synchronize()
    print(f"torch.bmm fp32, one two-pair 256x256 job: {(time.perf_counter() - t0) * 100:.3f} ms  (x ~10 for a step)")

if __name__ == "__main__":
    main()
