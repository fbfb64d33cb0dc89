#!/usr/bin/env python3
"""Hypothesis test for the f16x3 gradient error at 1 024 rays (profiles/r05/train1024_diag.log): the adjoint chains carry values
~ 1 / N; if fp16 underflow of their hi / lo halves is the cause, a loss scaled by 2^k before backward() (and divided out of the
gradients afterwards - every backward kernel is linear in its seeds) brings the error down to the f32 mode's."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nrhints_amd as na
from nrhints_amd.synthetic import perturb_state
from nrhints_amd.training import train_loss_dict
from tests.conftest import load_npz

T = torch.from_numpy
cu = lambda a: T(np.asarray(a)).float().contiguous().cuda()
g = load_npz("train1024_b.npz")
state = perturb_state(load_npz("scene_a_state.npz"))
gs = 25000
p = f"s{gs}."
for prec, scale in (("f32", 1.0), ("f16x3", 1.0), ("f16x3", 64.0), ("f16x3", 1024.0), ("f16x3", 32768.0)):
    m = na.NeuSHintRenderer(na.NeuSModelConfig(), precision=prec)
    m.load_state_dict({k: T(np.asarray(v)) for k, v in state.items()})
    m = m.cuda()
    rb = na.RayBundle(origins=cu(g["o"]), directions=cu(g["d"]), pl_positions=cu(g["pl"]), nears=cu(g["near"]), fars=cu(g["far"]))
    out = m(rb, is_training=True, background_rgb=torch.ones(1, 3).cuda(), global_step=gs,
            _t_rand_primary=cu(g[p + "t_rand_primary"]), _t_rand_shadow=cu(g[p + "t_rand_shadow"]))
    ld = train_loss_dict(out, cu(g["rgb_gt"]), m.config.igr_weight)
    (ld["loss"] * scale).backward()
    rows = []
    for name, prm in m.named_parameters():
        want = g[p + "grad64." + name].astype(np.float64)
        got = prm.grad.detach().cpu().numpy().astype(np.float64) / scale
        sc = max(np.abs(want).max(), 1e-12)
        noise = float(g[p + "noise." + name])
        emax = np.abs(got - want).max()
        rows.append((emax / max(3 * noise, 1e-4 * sc), name, emax / sc, noise / sc))
    rows.sort(reverse=True)
    print(f"== {prec} loss scale {scale:g}: over bound {sum(r[0] >= 1 for r in rows)}; top: " + "; ".join(f"{r[1].replace('_network', '')} {r[0]:.2f} ({r[2]:.1e})" for r in rows[:5]))
