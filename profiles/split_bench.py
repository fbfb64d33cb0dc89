#!/usr/bin/env python3
"""Per-pass latency of the SDF value kernels on small point sets: the wide evaluation kernel (sdf32_kernel<0>), the 16-point f16x3
kernel and the channel-split kernel (csrc/nrh_sdf_split.hip) with 1 / 2 tiles per workgroup.  Back-to-back launches on one
stream (each waits for the previous one, as the sampler's passes do); microseconds per pass."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nrhints_amd as na
from nrhints_amd import ops
from nrhints_amd.synthetic import make_rays


def timeit(fn, reps=40):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    m = na.NeuSHintRenderer(precision="f16x3").cuda().eval()
    p = m.packed_params(torch.device("cuda", 0))
    print(f"{'rays x per':>12s} {'points':>7s} {'wide':>8s} {'16pt':>8s} {'split T1':>9s} {'split T2':>9s}   (us per pass)")
    for nrays, nper in ((64, 16), (128, 16), (64, 64), (256, 16), (128, 64), (512, 16), (256, 64), (1024, 16), (512, 64), (1024, 64)):
        o, d, pl, near, far = (torch.from_numpy(a).cuda() for a in make_rays(nrays, seed=3, spread=0.1))
        t = (near + (far - near) * torch.linspace(0, 1, nper, device="cuda")[None]).contiguous()
        a16 = (p["sdf_w"], p["sdf_b"], p["sdf_head"], o, d, t, nper)
        r = [timeit(lambda: ops.sdf_eval_wide(0, p["sdf_w32"], p["sdf_tab32"], o, d, t, nper)),
             timeit(lambda: ops.sdf_eval(0, *a16)),
             timeit(lambda: ops.sdf_eval_split(*a16, tiles=1)),
             timeit(lambda: ops.sdf_eval_split(*a16, tiles=2))]
        print(f"{nrays:>7d} x {nper:<3d} {nrays * nper:>7d} " + " ".join(f"{x:8.1f}" for x in r), flush=True)


if __name__ == "__main__":
    main()
