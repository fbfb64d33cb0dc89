#!/usr/bin/env python3
"""Where does a GraphedTrainStep replay differ from the eager step?  Per tensor: gradient and parameter after step 0/1."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import nrhints_amd as na
from nrhints_amd.synthetic import make_rays, perturb_state
from nrhints_amd.training import GraphedTrainStep, lr_factor, train_loss_dict

T = torch.from_numpy
st = perturb_state(dict(np.load(os.path.join(os.path.dirname(__file__), "../../tests/golden/scene_a_state.npz"))))
def model():
    m = na.NeuSHintRenderer(na.NeuSModelConfig(), precision="f16x3")
    m.load_state_dict({k: T(np.asarray(v)) for k, v in st.items()})
    return m.cuda()
n, lr, gs = 128, 5e-4, 30000
bg = torch.ones(1, 3).cuda()
rs = np.random.RandomState(5)
cu = lambda a: T(a).float().contiguous().cuda()
def bundle(i):
    o, d, pl, near, far = make_rays(n, seed=40 + i, spread=0.1)
    return na.RayBundle(origins=cu(o), directions=cu(d), pl_positions=cu(pl), nears=cu(near), fars=cu(far))
batches = [(bundle(i), cu(rs.rand(n, 3).astype(np.float32))) for i in range(2)]
jit = [(cu(rs.rand(n, 1).astype(np.float32)), cu(rs.rand(n, 64).astype(np.float32))) for _ in range(2)]
e = model()
for capturable in (False, True):
    e = model()
    opt = torch.optim.Adam([{"params": list(e.parameters()), "lr": torch.tensor(lr, device="cuda") if capturable else lr}], capturable=capturable)
    rec = []
    for i in range(2):
        for grp in opt.param_groups:
            if capturable: grp["lr"].fill_(lr * lr_factor(gs + i, 20, 1_000_000, 0.05))
            else: grp["lr"] = lr * lr_factor(gs + i, 20, 1_000_000, 0.05)
        out = e(batches[i][0], is_training=True, background_rgb=bg, global_step=gs + i, _t_rand_primary=jit[i][0], _t_rand_shadow=jit[i][1])
        ld = train_loss_dict(out, batches[i][1], e.config.igr_weight)
        opt.zero_grad(set_to_none=True)
        ld["loss"].backward()
        g = {k: p.grad.detach().clone() for k, p in e.named_parameters()}
        opt.step()
        rec.append((float(ld["loss"].detach()), g, {k: p.detach().clone() for k, p in e.named_parameters()}))
    if not capturable: rec_e = rec
    else: rec_c = rec
gm = model()
step = GraphedTrainStep(gm, n, bg, lr=lr, warm_up_end=20, global_step=gs, jitter=(torch.zeros(n, 1), torch.zeros(n, 64)))
for i in range(2):
    step.jitter[0].copy_(jit[i][0]); step.jitter[1].copy_(jit[i][1])
    loss = step(batches[i][0], batches[i][1], global_step=gs + i)["loss"]
    print(f"step {i}: loss eager {rec_e[i][0]:.9f} eager-capturable {rec_c[i][0]:.9f} graph {loss:.9f}")
    worst = []
    for k, p in gm.named_parameters():
        ge, pe = rec_e[i][1][k], rec_e[i][2][k]
        dg = float((p.grad - ge).abs().max() / (ge.abs().max() + 1e-30))
        dp = float((p.detach() - pe).abs().max())
        dpc = float((rec_c[i][2][k] - pe).abs().max())
        worst.append((dp, k, dg, dpc))
    for dp, k, dg, dpc in sorted(worst, reverse=True)[:6]:
        print(f"   {k:40s} max|dparam| graph-eager {dp:.3e}  (capturable-eager {dpc:.3e})   rel max|dgrad| {dg:.3e}")
