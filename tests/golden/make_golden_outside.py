#!/usr/bin/env python3
"""Golden fixture for the outside-NeRF background (renderer.use_outside_nerf), recorded by IMPORTING the
reference (models/neus_hint_model.py:260-266, :434-473, :516-519, :630-633, :677-724; fields/nerf_density_field.py).  Build
container only; writes tests/golden/outside_b.npz (plain data).

    python tests/golden/make_golden_outside.py

Scene b's renderer weights + the reference's own initialisation of the NeRF (torch.manual_seed(0) constructor stream), with the
density head's bias raised so that the background actually contributes on these rays.  Recorded: the NeRF's state dict, its unit
I/O, one evaluation render and one training step (loss, a subset of the gradients in float32 and float64, the three jitter draws).
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

KEEP_GRADS = ("color_network.lin0.weight_v", "color_network.lin2.bias", "sdf_network.lin0.weight_v", "sdf_network.lin4.weight_g",
              "sdf_network.lin7.bias", "sdf_network.out_feat.weight_g", "deviation_network.variance", "outside_nerf.pts_linears.0.weight",
              "outside_nerf.pts_linears.5.weight", "outside_nerf.pts_linears.7.bias", "outside_nerf.alpha_linear.weight",
              "outside_nerf.feature_linear.weight", "outside_nerf.views_linears.0.weight", "outside_nerf.rgb_linear.bias")


def _install_stubs():
    class _Sub:
        def __getitem__(self, item):
            return object

    jt = types.ModuleType("jaxtyping")
    for name in ("Float", "Int", "Shaped", "Bool"):
        setattr(jt, name, _Sub())
    sys.modules["jaxtyping"] = jt
    sys.modules["mcubes"] = types.ModuleType("mcubes")


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)
    import torch

    torch.set_num_threads(8)
    from camera.ray_utils import RayBundle
    from models.neus_hint_model import NeuSHintRenderer, NeuSModelConfig, NeuSRendererConfig

    from nrhints_amd.synthetic import make_rays, perturb_state

    state_b = perturb_state(dict(np.load(os.path.join(HERE, "scene_a_state.npz"))))
    cfg = NeuSModelConfig(renderer=NeuSRendererConfig(use_outside_nerf=True))

    def build(dtype=torch.float32):
        torch.manual_seed(0)
        m = NeuSHintRenderer(cfg)
        sd = m.state_dict()
        sd.update({k: torch.from_numpy(np.asarray(v)) for k, v in state_b.items()})
        sd["outside_nerf.alpha_linear.bias"] = sd["outside_nerf.alpha_linear.bias"] + 1.5     # visible background density
        m.load_state_dict(sd)
        return m.to(dtype)

    def bundle(arrs, dtype=torch.float32, grad=False):
        ts = [torch.from_numpy(a).to(dtype).clone().requires_grad_(grad and i < 3) for i, a in enumerate(arrs)]
        return RayBundle(origins=ts[0], directions=ts[1], pl_positions=ts[2], nears=ts[3], fars=ts[4]), ts

    m = build().eval()
    rec = {"nerf." + k[len("outside_nerf."):]: v.numpy().copy() for k, v in m.state_dict().items() if k.startswith("outside_nerf.")}
    # unit I/O of the NeRF (fields/nerf_density_field.py:66-89)
    g = torch.Generator().manual_seed(3)
    x = torch.nn.functional.normalize(torch.randn(96, 3, generator=g), dim=-1)
    inv = torch.rand(96, 1, generator=g)
    pts4 = torch.cat([x, inv], dim=-1)
    views = torch.nn.functional.normalize(torch.randn(96, 3, generator=g), dim=-1)
    pls = torch.randn(96, 3, generator=g) * 3
    with torch.no_grad():
        dens, col = m.outside_nerf(pts4, views, pls)
    rec.update({"unit.pts4": pts4.numpy(), "unit.views": views.numpy(), "unit.pls": pls.numpy(), "unit.density": dens.numpy(), "unit.rgb": col.numpy()})

    N = 64
    rays = make_rays(N, seed=41, spread=0.25)      # wide spread: a good share of the rays misses the object and sees the background
    rec.update(dict(zip(("o", "d", "pl", "near", "far"), rays)))
    rb, _ = bundle(rays)
    with torch.no_grad():
        r = m(rb, is_training=False, background_rgb=torch.ones(1, 3))
    for name in ("rgb", "depth", "weights", "visibilities", "specular_cue", "inside_sphere", "normalized_analytic_normals"):
        rec["eval." + name] = getattr(r, name).detach().numpy()
    print("eval rgb mean", float(r.rgb.mean()), "weights shape", tuple(r.weights.shape), "tail weight mean", float(r.weights[:, 128:].sum(-1).mean()))

    Nt = 32
    trays = make_rays(Nt, seed=43, spread=0.25)
    rec.update({"t." + k: v for k, v in zip(("o", "d", "pl", "near", "far"), trays)})
    gt = torch.full((Nt, 3), 0.5)
    rec["t.rgb_gt"], rec["t.global_step"] = gt.numpy(), np.int64(20000)
    real_rand = torch.rand
    drawn = []

    def rec_rand(*a, **k):
        t = real_rand(*a, **k)
        drawn.append(t.detach().clone())
        return t

    for dt, sfx in ((torch.float32, ""), (torch.float64, "64")):
        mm = build(dt).train()
        replay = [x.to(dt) for x in drawn]
        torch.manual_seed(5)
        torch.rand = rec_rand if dt == torch.float32 else (lambda *a, **k: replay.pop(0))
        try:
            rbt, ts = bundle(trays, dt, grad=True)
            r = mm(rbt, is_training=True, background_rgb=torch.ones(1, 3, dtype=dt), global_step=20000)
        finally:
            torch.rand = real_rand
        g_ = gt.to(dt)
        rgb_loss = torch.nn.functional.l1_loss(r.rgb, g_, reduction="sum") / (Nt + 1e-5)
        ge = (torch.linalg.norm(r.analytic_normals, ord=2, dim=-1) - 1.0) ** 2
        eik = (r.relax_inside_sphere * ge).sum() / (r.relax_inside_sphere.sum() + 1e-5)
        loss = rgb_loss + 0.1 * eik
        loss.backward()
        if dt == torch.float32:
            assert [tuple(t.shape) for t in drawn] == [(Nt, 1), (Nt, 32), (Nt, 64)], [tuple(t.shape) for t in drawn]
            rec["t.t_rand_primary"], rec["t.t_rand_outside"], rec["t.t_rand_shadow"] = (t.numpy() for t in drawn)
            rec["t.rgb"], rec["t.weights"] = r.rgb.detach().numpy(), r.weights.detach().numpy()
        rec[f"t.loss{sfx}"] = loss.detach().numpy()
        for name, prm in mm.named_parameters():
            if name in KEEP_GRADS:
                rec[f"t.grad{sfx}.{name}"] = prm.grad.detach().numpy().copy()
        for nm, t in zip(("origins", "directions", "pl_positions"), ts):
            rec[f"t.grad{sfx}.rays.{nm}"] = t.grad.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "outside_b.npz"), **rec)
    print({k: v.shape for k, v in rec.items() if k.startswith("nerf.")})


if __name__ == "__main__":
    main()
