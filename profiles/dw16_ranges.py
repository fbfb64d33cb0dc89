#!/usr/bin/env python3
"""Ranges of the weight-gradient kernel's operand arrays (max |x| per array, in binades) over scenes / anneal steps / batch sizes:
what a STATIC fp16 scale per array class has to cover (profiles/dw16_emulation.py showed fp16 operands with the array maximum
placed at 2^10 stay inside every gradient bound; the usable window for max * scale is [2^-13, 2^15])."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nrhints_amd as na  # noqa: E402
from nrhints_amd import dw, train_fused  # noqa: E402
from nrhints_amd.synthetic import make_rays, perturb_state  # noqa: E402
from tests.conftest import load_npz  # noqa: E402

T = torch.from_numpy
cu = lambda a: T(np.asarray(a)).float().contiguous().cuda()  # noqa: E731
real_run = dw.run
LOG = {}
CUR = {"k": None}
NAMES = ["sdf0"] + [f"sdf{l}" for l in range(1, 8)] + ["feat_head", "sdf_head", "col0f", "col0m", "col1", "col2", "col3", "col4"]


def spy(jobs, npts, total_items=None):
    for name, j in zip(NAMES, jobs):
        for side, arrs in (("A", j.a), ("B", j.b)):
            for k, x in enumerate(arrs):
                mx = float(x.abs().max())
                LOG.setdefault((name, side, k), {})[CUR["k"]] = mx
    return real_run(jobs, npts, total_items)


dw.run = spy
a_state = {k: np.asarray(v) for k, v in load_npz("scene_a_state.npz").items()}
for scene, state in (("a", a_state), ("b", perturb_state(dict(a_state)))):
    for n in (1024, 64):
        for gs in (0, 100000):
            CUR["k"] = (scene, n, gs)
            m = na.NeuSHintRenderer(na.NeuSModelConfig(), precision="f16x3")
            m.load_state_dict({k: T(np.asarray(v)) for k, v in state.items()})
            m = m.cuda()
            o, d, pl, near, far = make_rays(n, seed=41, spread=0.1)
            rb = na.RayBundle(origins=cu(o), directions=cu(d), pl_positions=cu(pl), nears=cu(near), fars=cu(far))
            gt = cu(np.random.RandomState(n).rand(n, 3).astype(np.float32))
            train_fused.train_step_backward(m, rb, gt, torch.ones(1, 3).cuda(), gs)
keys = sorted({k for v in LOG.values() for k in v})
print("log2(max |x| * rays) per operand array; columns:", keys)
for (name, side, k), v in LOG.items():
    print(f"{name:10s} {side}{k}  " + "  ".join(f"{np.log2(max(v[c], 1e-300) * c[1]):7.1f}" for c in keys))
