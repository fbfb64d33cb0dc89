#!/usr/bin/env python3
"""nrh_dw_gemm on the seven two-pair 256 x 256 products of the SDF net at 1 024 rays (131 072 points): float32 tiled operands
(three bf16 MFMA passes) against float16 half-tiled operands (one fp16 pass, NrhDwJob.half_ops).  HIP events over 20 calls."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nrhints_amd import dw  # noqa: E402

P = 131072
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
arr32 = [torch.randn(8, P, 256, device=dev, generator=g) for _ in range(4)]          # zbar, t, h, abar
arr16 = [a.half() for a in arr32]
out = [torch.empty(256, 256, device=dev) for _ in range(7)]
db = [torch.empty(256, device=dev) for _ in range(7)]
dyn = torch.tensor([1.0, 1.0], device=dev)


def jobs(half):
    z, t, h, ab = arr16 if half else arr32
    if half:
        return [dw.Job([z[l], t[l]], [h[l - 1], ab[l - 1]], 256, 256, out[l - 1], colsum_a=db[l - 1], half=True, dyn_scale=dyn) for l in range(1, 8)]
    return [dw.Job([z[l], t[l]], [h[l - 1], ab[l - 1]], 256, 256, out[l - 1], colsum_a=db[l - 1], tiled_a=(True, True), tiled_b=(True, True))
            for l in range(1, 8)]


for half in (False, True, False, True):
    js = jobs(half)
    for _ in range(3):
        dw.run(js, P)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        dw.run(js, P)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    gb = 7 * 4 * P * 256 * (2 if half else 4) / 1e9
    print(f"{'float16 half-tiled, 1 pass ' if half else 'float32 tiled, bf16 x 3    '}: {ms:.3f} ms per call, {gb:.2f} GB of operands -> {gb / ms:.2f} TB/s", flush=True)
