#!/usr/bin/env python3
"""Diagnosis aid for tests/test_gpu_train1024.py: per tensor, |ours - ref64| against the reference's own |ref32 - ref64|
(max norm, and L2 when the fixture carries it), for the fused step in both precisions at the three global steps."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nrhints_amd as na
from nrhints_amd import train_fused
from nrhints_amd.synthetic import perturb_state
from tests.conftest import load_npz

T = torch.from_numpy
cu = lambda a: T(np.asarray(a)).float().contiguous().cuda()
g = load_npz("train1024_b.npz")
state = perturb_state(load_npz("scene_a_state.npz"))
res = {}
for prec in ("f16x3", "f32"):
    for gs in (0, 25000, 100000):
        p = f"s{gs}."
        m = na.NeuSHintRenderer(na.NeuSModelConfig(), precision=prec)
        m.load_state_dict({k: T(np.asarray(v)) for k, v in state.items()})
        m = m.cuda()
        rb = na.RayBundle(origins=cu(g["o"]), directions=cu(g["d"]), pl_positions=cu(g["pl"]), nears=cu(g["near"]), fars=cu(g["far"]))
        l8 = train_fused.train_step_backward(m, rb, cu(g["rgb_gt"]), torch.ones(1, 3).cuda(), gs, t_rand_primary=cu(g[p + "t_rand_primary"]),
                                             t_rand_shadow=cu(g[p + "t_rand_shadow"]))
        ld = train_fused.loss_dict(l8)
        rows = []
        for name, prm in m.named_parameters():
            want = g[p + "grad64." + name].astype(np.float64)
            got = prm.grad.detach().cpu().numpy().astype(np.float64)
            res[(prec, gs, name)] = got
            scale = max(np.abs(want).max(), 1e-12)
            noise = float(g[p + "noise." + name])
            emax = np.abs(got - want).max()
            e2 = np.sqrt(((got - want) ** 2).sum())
            n2 = float(g[p + "noise2." + name]) if (p + "noise2." + name) in g else float("nan")
            rows.append((emax / max(3 * noise, 1e-4 * scale), name, emax / scale, noise / scale, e2 / max(n2, 1e-300), want.size))
        rows.sort(reverse=True)
        print(f"== {prec} step {gs}: loss {ld['loss']:.6f} (ref64 {float(g[p + 'loss_f64']):.6f}, ref32 {float(g[p + 'loss']):.6f}); tensors over bound: {sum(r[0] >= 1 for r in rows)}")
        for r in rows[:8]:
            print(f"   ratio {r[0]:5.2f}  {r[1]:38s} err/scale {r[2]:.2e}  refnoise/scale {r[3]:.2e}  L2 err/refnoise {r[4]:.2f}  n={r[5]}")
# our two precisions against each other
for gs in (0, 25000, 100000):
    worst = max((np.abs(res[("f16x3", gs, n)] - res[("f32", gs, n)]).max() / max(np.abs(res[("f32", gs, n)]).max(), 1e-12), n)
                for (pr, s_, n) in res if pr == "f32" and s_ == gs)
    print(f"f16x3 vs f32 at step {gs}: worst |diff| / scale = {worst[0]:.2e} on {worst[1]}")
