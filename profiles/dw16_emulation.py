#!/usr/bin/env python3
"""Go / no-go for 16-bit operands of the weight-gradient kernel (CHANGELOG.md section 7c item 3: "16-bit storage of the dW-only operands would
halve four units and is not float32-equivalent") - priced on the GPU before anything is built: the fused 1 024-ray step of
tests/test_gpu_train1024.py with the operand arrays of nrh_dw_gemm ROUNDED IN PLACE (torch) right before the call, every tensor's
gradient against the reference's float64 gradient in units of the test's bound (3 x the reference's float32 noise, floor 1e-4 of scale;
"pooled" = the f16x3 test's yardstick, the largest of the reference's three draws).

modes:  none      the shipped path (float32 arrays, bf16x3 products)
        bf16      every operand rounded to bf16 (what a single v_mfma_f32_32x32x16_bf16 per K step would see)
        fp16      activations (h, t, feature, inputs) to fp16; adjoints (zbar, abar, ...) scaled per ARRAY by a power of two that puts
                  the array's maximum at 2^10, rounded to fp16, scaled back (the best a static scale could do)
        fp16fix   as fp16, the adjoints' scale fixed to 2^14 x adjoint_scale(rays) for every array (a scale the kernels could apply
                  without looking at the data); reports how many elements overflow / flush to zero
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nrhints_amd as na  # noqa: E402
from nrhints_amd import _lib, dw, train_fused  # noqa: E402
from nrhints_amd.synthetic import perturb_state  # noqa: E402
from tests.conftest import load_npz  # noqa: E402

T = torch.from_numpy
cu = lambda a: T(np.asarray(a)).float().contiguous().cuda()  # noqa: E731
g = load_npz("train1024_b.npz")
state = perturb_state(load_npz("scene_a_state.npz"))
STEPS = (0, 25000, 100000)
MODE = {"m": "none"}
STATS = {}
real_run = dw.run


def _is_activation(x):
    # activations are O(1..100); adjoints of a mean loss over 1 024 rays are < 1e-2
    return float(x.abs().max()) > 0.05


def _round(x, mode, tag):
    if mode == "bf16":
        x.copy_(x.to(torch.bfloat16).float())
        return
    mx = float(x.abs().max())
    if mx == 0.0:
        return
    if _is_activation(x):
        s = 1.0
    elif mode == "fp16":
        s = 2.0 ** (10 - int(np.ceil(np.log2(mx))))
    else:
        s = 2.0 ** 14 * _lib.adjoint_scale(1024)
    y = (x * s)
    over = int((y.abs() > 65504).sum())
    flush = int(((y.abs() < 2.0 ** -25) & (x != 0)).sum())
    st = STATS.setdefault(mode, dict(over=0, flush=0, n=0, max_scaled=0.0))
    st["over"] += over
    st["flush"] += flush
    st["n"] += x.numel()
    if s != 1.0:
        st["max_scaled"] = max(st["max_scaled"], mx * s)
    x.copy_(y.half().float() / s)


def run_rounded(jobs, npts, total_items=None):
    if MODE["m"] != "none":
        seen = set()
        for j in jobs:
            for x in list(j.a) + list(j.b):
                if x.data_ptr() in seen:
                    continue
                seen.add(x.data_ptr())
                _round(x, MODE["m"], "")
    return real_run(jobs, npts, total_items)


dw.run = run_rounded


def bound(p, key, want, pooled):
    scale = max(float(np.abs(want).max()), 1e-12)
    if pooled:
        rel = max(float(g[f"s{s}.noise." + key]) / max(float(np.abs(g[f"s{s}.grad64." + key]).max()), 1e-12) for s in STEPS)
        noise = rel * scale
    else:
        noise = float(g[p + "noise." + key])
    return max(3 * noise, 1e-4 * scale), scale


for mode in ("none", "bf16", "fp16", "fp16fix"):
    MODE["m"] = mode
    for gs in STEPS:
        p = f"s{gs}."
        m = na.NeuSHintRenderer(na.NeuSModelConfig(), precision="f16x3")
        m.load_state_dict({k: T(np.asarray(v)) for k, v in state.items()})
        m = m.cuda()
        rb = na.RayBundle(origins=cu(g["o"]), directions=cu(g["d"]), pl_positions=cu(g["pl"]), nears=cu(g["near"]), fars=cu(g["far"]))
        train_fused.train_step_backward(m, rb, cu(g["rgb_gt"]), torch.ones(1, 3).cuda(), gs, t_rand_primary=cu(g[p + "t_rand_primary"]),
                                        t_rand_shadow=cu(g[p + "t_rand_shadow"]))
        rows = []
        for name, prm in m.named_parameters():
            want = g[p + "grad64." + name].astype(np.float64)
            got = prm.grad.detach().cpu().numpy().astype(np.float64)
            err = np.abs(got - want).max()
            b1, scale = bound(p, name, want, False)
            b3, _ = bound(p, name, want, True)
            rows.append((err / b3, err / b1, name, err / scale))
        rows.sort(reverse=True)
        print(f"== dW operands {mode:8s} step {gs:6d}: tensors over the pooled bound {sum(r[0] >= 1 for r in rows):2d}, over the per-step bound "
              f"{sum(r[1] >= 1 for r in rows):2d};  worst: " + "; ".join(f"{r[2]} {r[0]:.2f} ({r[3]:.1e} of scale)" for r in rows[:4]), flush=True)
for k, v in STATS.items():
    print(f"{k}: scaled adjoint maximum {v['max_scaled']:.3g}; overflowing elements {v['over']}, flushed to zero {v['flush']} of {v['n']}")
