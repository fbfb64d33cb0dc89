#!/usr/bin/env python3
"""The reference with network widths / encoding resolutions BELOW its defaults (fields/sdf_field.py:11-36,
fields/reflectance_network.py:9-22): evaluation renders and one training step per variant, float32 and float64 - fixtures for the
zero-padded execution of narrower networks on the compiled kernels (nrhints_amd/packing.py pad_to_compiled; VERDICT r5 missing #2).
Build container only (imports /root/reference); writes data:

    python tests/golden/make_golden_shapes.py      ->  tests/golden/render_shapes.npz

variants (sdf d_hidden / multi_res / d_out_feat | reflectance d_hidden / multi_res | hints):
  n128    128 / 4 / 128 | 128 / 2 | both      everything narrower; skip layer 101 rows, embedding 27 columns
  n192    192 / 6 / 64  | 256 / 4 | both      only the SDF width and the feature vector
  n160s   160 / 5 / 256 | 96 / 3  | shadow    a one-hint model of a narrower shape
Per variant the reference's own initialisation under torch.manual_seed(0), perturbed by nrhints_amd.synthetic.perturb_state
(pe_cols = 6 multi_res); recorded: the float64 sum of every tensor of the initialisation (the init RNG stream of the constructor,
from which the test rebuilds the state).  Rays make_rays(64, seed=37, spread=0.12) for evaluation, make_rays(32, seed=31, spread=0.1) for the training step
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
    "n128": (dict(d_hidden=128, multi_res=4, d_out_feat=128), dict(d_hidden=128, multi_res=2), dict()),
    "n192": (dict(d_hidden=192, multi_res=6, d_out_feat=64), dict(), dict()),
    "n160s": (dict(d_hidden=160, multi_res=5), dict(d_hidden=96, multi_res=3), dict(specular_hint=False)),
}


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)
    import torch
    torch.set_num_threads(8)
    from camera.ray_utils import RayBundle  # reference
    from fields.reflectance_network import ReflectanceNetConfig  # reference
    from fields.sdf_field import SDFNetConfig  # reference
    from models.neus_hint_model import NeuSHintRenderer, NeuSModelConfig, NeuSRendererConfig  # reference
    from nrhints_amd.synthetic import make_rays, perturb_state

    rays = make_rays(64, seed=37, spread=0.12)
    trays = make_rays(32, seed=31, spread=0.1)
    Nt, gs = 32, 20000
    gt = torch.full((Nt, 3), 0.5)
    rec = dict(zip(("o", "d", "pl", "near", "far"), rays))
    rec.update({"t." + k: v for k, v in zip(("o", "d", "pl", "near", "far"), trays)})
    rec["t.rgb_gt"], rec["t.global_step"] = gt.numpy(), np.int64(gs)
    real_rand = torch.rand
    for vt, (skw, ckw, rkw) in VARIANTS.items():
        def cfg():
            return NeuSModelConfig(sdf_network=SDFNetConfig(**skw), reflectance_network=ReflectanceNetConfig(**ckw),
                                   renderer=NeuSRendererConfig(**rkw))
        torch.manual_seed(0)
        init = {k: v.detach().numpy().copy() for k, v in NeuSHintRenderer(cfg()).state_dict().items()}
        for k, v in init.items():
            rec[f"{vt}.init_sum.{k}"] = np.float64(v.astype(np.float64).sum())
        state = perturb_state(init, pe_cols=6 * skw.get("multi_res", 6))
        # (the state itself is not stored: nrhints_amd's constructor reproduces the reference's init bit for bit under the same seed -
        # the test checks that against the sums above - and perturb_state is shared code)

        def build(dt):
            torch.manual_seed(0)
            m = NeuSHintRenderer(cfg())
            m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
            return m.to(dt)

        for dt, sfx in ((torch.float32, ""), (torch.float64, "_f64")):
            m = build(dt).eval()
            rb = RayBundle(origins=torch.from_numpy(rays[0]).to(dt), directions=torch.from_numpy(rays[1]).to(dt),
                           pl_positions=torch.from_numpy(rays[2]).to(dt), nears=torch.from_numpy(rays[3]).to(dt),
                           fars=torch.from_numpy(rays[4]).to(dt))
            with torch.no_grad():
                r = m(rb, is_training=False, background_rgb=torch.ones(1, 3, dtype=dt))
            names = ("rgb", "depth", "weights", "visibilities", "specular_cue", "inside_sphere", "normalized_analytic_normals", "s_val") \
                if dt == torch.float32 else ("rgb", "depth", "visibilities", "weights")
            for name in names:
                if getattr(r, name) is not None:
                    rec[f"{vt}.{name}{sfx}"] = getattr(r, name).detach().numpy()
        print(vt, "eval: rgb mean", float(rec[f"{vt}.rgb"].mean()), "rgb std", float(rec[f"{vt}.rgb"].std()),
              "max |rgb32 - rgb64|", float(np.abs(rec[f"{vt}.rgb"] - rec[f"{vt}.rgb_f64"]).max()),
              "weight sum mean", float(rec[f"{vt}.weights"].sum(1).mean()))
        drawn = []

        def rec_rand(*a, **k):
            t = real_rand(*a, **k)
            drawn.append(t.detach().clone())
            return t

        for dt, sfx in ((torch.float32, ""), (torch.float64, "64")):
            m = build(dt).train()
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
        print(vt, "train: loss", float(rec[f"{vt}.loss"]), float(rec[f"{vt}.loss64"]))
    np.savez_compressed(os.path.join(HERE, "render_shapes.npz"), **rec)


if __name__ == "__main__":
    main()
