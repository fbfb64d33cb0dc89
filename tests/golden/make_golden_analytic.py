#!/usr/bin/env python3
"""One training step of the reference with renderer.normal_type = Analytic (models/neus_hint_model.py:622-623: the reflectance net
reads the raw SDF gradient instead of its normalisation), jitter recorded, float32 and float64 - the fixture for the fused
(autograd-free) step's Analytic branch (round 5).  Build container only (imports /root/reference); writes data:

    python tests/golden/make_golden_analytic.py      ->  tests/golden/train_analytic_b.npz

Same rays (make_rays(32, seed=31, spread=0.1)), ground truth, global_step and recorded tensors as the one-hint steps of
make_golden_branches.py (KEEP_GRADS + the three ray gradients)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
from make_golden import REF, _install_stubs  # noqa: E402
from make_golden_branches import KEEP_GRADS  # noqa: E402


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)
    import torch
    torch.set_num_threads(8)
    from camera.ray_utils import RayBundle  # reference
    from models.neus_hint_model import NeuSHintRenderer, NeuSModelConfig, NeuSRendererConfig, NormalComputationType  # reference
    from nrhints_amd.synthetic import make_rays, perturb_state

    state_b = perturb_state(dict(np.load(os.path.join(HERE, "scene_a_state.npz"))))
    Nt, gs = 32, 20000
    trays = make_rays(Nt, seed=31, spread=0.1)
    gt = torch.full((Nt, 3), 0.5)
    rec = {"t." + k: v for k, v in zip(("o", "d", "pl", "near", "far"), trays)}
    rec["t.rgb_gt"], rec["t.global_step"] = gt.numpy(), np.int64(gs)
    real_rand = torch.rand
    drawn = []

    def rec_rand(*a, **k):
        t = real_rand(*a, **k)
        drawn.append(t.detach().clone())
        return t

    for dt, sfx in ((torch.float32, ""), (torch.float64, "64")):
        torch.manual_seed(0)
        m = NeuSHintRenderer(NeuSModelConfig(renderer=NeuSRendererConfig(normal_type=NormalComputationType.Analytic)))
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state_b.items()})
        m = m.to(dt).train()
        replay = [x.to(dt) for x in drawn]
        torch.manual_seed(5)
        torch.rand = rec_rand if dt == torch.float32 else (lambda *a, **k: replay.pop(0))
        try:
            ts = [torch.from_numpy(a).to(dt).clone().requires_grad_(i < 3) for i, a in enumerate(trays)]
            rb = RayBundle(origins=ts[0], directions=ts[1], pl_positions=ts[2], nears=ts[3], fars=ts[4])
            r = m(rb, is_training=True, background_rgb=torch.ones(1, 3, dtype=dt), global_step=gs)
        finally:
            torch.rand = real_rand
        g = gt.to(dt)
        rgb_loss = torch.nn.functional.l1_loss(r.rgb, g, reduction="sum") / (Nt + 1e-5)
        ge = (torch.linalg.norm(r.analytic_normals, ord=2, dim=-1) - 1.0) ** 2
        eik = (r.relax_inside_sphere * ge).sum() / (r.relax_inside_sphere.sum() + 1e-5)
        loss = rgb_loss + 0.1 * eik
        loss.backward()
        if dt == torch.float32:
            assert len(drawn) == 2
            rec["ana.t_rand_primary"], rec["ana.t_rand_shadow"] = drawn[0].numpy(), drawn[1].numpy()
            rec["ana.t.rgb"] = r.rgb.detach().numpy()
        rec[f"ana.loss{sfx}"] = loss.detach().numpy()
        for name, prm in m.named_parameters():
            if name in KEEP_GRADS:
                rec[f"ana.grad{sfx}.{name}"] = prm.grad.detach().numpy().copy()
        for nm, t in zip(("origins", "directions", "pl_positions"), ts):
            rec[f"ana.grad{sfx}.rays.{nm}"] = t.grad.detach().numpy().copy()
        print("analytic step", dt, "loss", float(loss))
    np.savez_compressed(os.path.join(HERE, "train_analytic_b.npz"), **rec)


if __name__ == "__main__":
    main()
