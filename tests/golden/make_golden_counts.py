#!/usr/bin/env python3
"""The reference at sample counts off its defaults (models/neus_hint_model.py:139-171, :696-713, :373-412): evaluation renders and
one training step per variant, float32 and float64 - fixtures for the sample-count generality of the kernels.
Build container only (imports /root/reference); writes data:

    python tests/golden/make_golden_counts.py      ->  tests/golden/render_counts_b.npz

variants (n_samples + n_importance_samples / up_sample_steps | n_shadow_samples + n_shadow_importance_samples):
  c3232   32 + 32 / 2  | 64 + 64     64 samples per ray, 16 new per step
  c6432   64 + 32 / 2  | 64 + 64     96 samples per ray
  c4848   48 + 48 / 4  | 32 + 32     96 samples per ray, 12 new per step; 64 on the shadow ray, 8 new per step
  c8000   80 + 0       | 48 + 0      no hierarchical sampling on either ray; more than 64 coarse samples
Scene b; rays make_rays(64, seed=37, spread=0.12) for evaluation, make_rays(32, seed=31, spread=0.1) for the training step
(global_step 20 000, ground truth 0.5, recorded jitter; KEEP_GRADS of make_golden_branches.py + the three ray gradients)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
from make_golden import REF, _install_stubs  # noqa: E402
from make_golden_branches import KEEP_GRADS  # noqa: E402

VARIANTS = {
    "c3232": dict(n_samples=32, n_importance_samples=32, up_sample_steps=2),
    "c6432": dict(n_samples=64, n_importance_samples=32, up_sample_steps=2),
    "c4848": dict(n_samples=48, n_importance_samples=48, up_sample_steps=4, n_shadow_samples=32, n_shadow_importance_samples=32),
    "c8000": dict(n_samples=80, n_importance_samples=0, n_shadow_samples=48, n_shadow_importance_samples=0),
}


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)
    import torch
    torch.set_num_threads(8)
    from camera.ray_utils import RayBundle  # reference
    from models.neus_hint_model import NeuSHintRenderer, NeuSModelConfig, NeuSRendererConfig  # reference
    from nrhints_amd.synthetic import make_rays, perturb_state

    state_b = perturb_state(dict(np.load(os.path.join(HERE, "scene_a_state.npz"))))

    def build(kw, dt):
        torch.manual_seed(0)
        m = NeuSHintRenderer(NeuSModelConfig(renderer=NeuSRendererConfig(**kw)))
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state_b.items()})
        return m.to(dt)

    rays = make_rays(64, seed=37, spread=0.12)
    trays = make_rays(32, seed=31, spread=0.1)
    Nt, gs = 32, 20000
    gt = torch.full((Nt, 3), 0.5)
    rec = dict(zip(("o", "d", "pl", "near", "far"), rays))
    rec.update({"t." + k: v for k, v in zip(("o", "d", "pl", "near", "far"), trays)})
    rec["t.rgb_gt"], rec["t.global_step"] = gt.numpy(), np.int64(gs)
    real_rand = torch.rand
    for vt, kw in VARIANTS.items():
        for dt, sfx in ((torch.float32, ""), (torch.float64, "_f64")):
            m = build(kw, dt).eval()
            rb = RayBundle(*[torch.from_numpy(a).to(dt) for a in rays][:3], nears=torch.from_numpy(rays[3]).to(dt), fars=torch.from_numpy(rays[4]).to(dt)) \
                if False else RayBundle(origins=torch.from_numpy(rays[0]).to(dt), directions=torch.from_numpy(rays[1]).to(dt),
                                        pl_positions=torch.from_numpy(rays[2]).to(dt), nears=torch.from_numpy(rays[3]).to(dt),
                                        fars=torch.from_numpy(rays[4]).to(dt))
            with torch.no_grad():
                r = m(rb, is_training=False, background_rgb=torch.ones(1, 3, dtype=dt))
            names = ("rgb", "depth", "weights", "visibilities", "specular_cue", "inside_sphere", "normalized_analytic_normals", "s_val") \
                if dt == torch.float32 else ("rgb", "depth", "visibilities", "weights")
            for name in names:
                rec[f"{vt}.{name}{sfx}"] = getattr(r, name).detach().numpy()
        print(vt, "eval: samples per ray", rec[f"{vt}.weights"].shape[1], "rgb mean", float(rec[f"{vt}.rgb"].mean()),
              "max |rgb32 - rgb64|", float(np.abs(rec[f"{vt}.rgb"] - rec[f"{vt}.rgb_f64"]).max()))
        drawn = []

        def rec_rand(*a, **k):
            t = real_rand(*a, **k)
            drawn.append(t.detach().clone())
            return t

        for dt, sfx in ((torch.float32, ""), (torch.float64, "64")):
            m = build(kw, dt).train()
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
                assert len(drawn) == 2, len(drawn)
                rec[f"{vt}.t_rand_primary"], rec[f"{vt}.t_rand_shadow"] = drawn[0].numpy(), drawn[1].numpy()
                rec[f"{vt}.t.rgb"] = r.rgb.detach().numpy()
            rec[f"{vt}.loss{sfx}"] = loss.detach().numpy()
            for name, prm in m.named_parameters():
                if name in KEEP_GRADS:
                    rec[f"{vt}.grad{sfx}.{name}"] = prm.grad.detach().numpy().copy()
            for nm, t in zip(("origins", "directions", "pl_positions"), ts):
                rec[f"{vt}.grad{sfx}.rays.{nm}"] = t.grad.detach().numpy().copy()
        print(vt, "train: loss", float(rec[f"{vt}.loss"]), "shadow jitter", rec[f"{vt}.t_rand_shadow"].shape)
    np.savez_compressed(os.path.join(HERE, "render_counts_b.npz"), **rec)


if __name__ == "__main__":
    main()
