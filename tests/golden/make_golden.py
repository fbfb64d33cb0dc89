#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by IMPORTING the reference.

Runs ONLY in the build container (needs /root/reference). Nothing on the GPU box
or in the product path may import this file; the fixtures it writes are plain
data (inputs + expected outputs), never reference source.

    python tests/golden/make_golden.py

What it records (all float32 unless suffixed _f64):

  scene_a_state.npz     state_dict of NeuSHintRenderer(NeuSModelConfig()) built
                        under torch.manual_seed(0)          (reference init,
                        models/neus_hint_model.py:237-267, fields/sdf_field.py:58-101)
  unit_<scene>.npz      unit-level I/O of the reference functions on the hot
                        path
  render_<scene>.npz    NeuSHintRenderer.forward(is_training=False) end to end,
                        fp32 and fp64 (row a12)
  train_<scene>.npz     forward(is_training=True) with the drawn jitter recorded

Scenes: "a" = reference init, variance 0.3; "b" = scene a with the deterministic
perturbation of nrhints_amd.synthetic.perturb_state (breaks the sphere symmetry, makes
every weight entry non-zero) and variance 0.7 (inv_s ~ 1.1e3, trained-like).
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def _install_stubs():
    """jaxtyping is annotation-only, mcubes is only used by extract_geometry."""

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
    from models.neus_hint_model import NeuSHintRenderer, NeuSModelConfig  # reference
    from camera.ray_utils import RayBundle  # reference

    from nrhints_amd.synthetic import make_rays, naive_state, perturb_state  # ours (pure numpy data helpers)
    from models.neus_hint_model import (NeuSRendererConfig, DepthComputationType, NormalComputationType)  # reference

    def build(state=None, dtype=torch.float32):
        torch.manual_seed(0)
        m = NeuSHintRenderer(NeuSModelConfig())
        if state is not None:
            m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
        return m.to(dtype).eval()

    def state_np(m):
        return {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}

    base = build()
    state_a = state_np(base)
    np.savez(os.path.join(HERE, "scene_a_state.npz"), **state_a)
    state_b = perturb_state(state_a)
    scenes = {"a": state_a, "b": state_b}

    for tag, state in scenes.items():
        m32 = build(state, torch.float32)
        m64 = build(state, torch.float64)

        # ---------------- unit-level records ----------------
        unit = {}
        g = torch.Generator().manual_seed(1234)
        pts = (torch.rand(192, 3, generator=g) * 2 - 1) * 0.9
        with torch.no_grad():
            out = m32.sdf_network(pts)
        unit["sdf_pts"] = pts.numpy()
        unit["sdf_out"] = out.numpy()  # [P,257] = [sdf, feat]
        with torch.no_grad():
            unit["sdf_out_f64"] = m64.sdf_network(pts.double()).numpy()
        grad = m32.sdf_network.gradient(pts.clone()).detach().squeeze(1)
        unit["sdf_grad"] = grad.numpy()
        unit["sdf_grad_f64"] = m64.sdf_network.gradient(pts.double().clone()).detach().squeeze(1).numpy()

        # NeRFEncoding on its own (fields/encodings.py:155-176)
        x = torch.randn(16, 3, generator=g)
        unit["enc_x"] = x.numpy()
        unit["enc6"] = m32.sdf_network.embed_fn_fine(x).numpy()
        unit["enc4"] = m32.color_network.embed_view_pl_fn(x).numpy()

        # sampler pieces on real rays
        N = 48
        o, d, pl, near, far = [torch.from_numpy(a) for a in make_rays(N, seed=7, spread=0.12)]
        with torch.no_grad():
            z = near + (far - near) * torch.linspace(0.0, 1.0, 64)[None, :]
            p = o[:, None, :] + d[:, None, :] * z[..., None]
            sdf0 = m32.sdf_network.sdf(p.reshape(-1, 3)).reshape(N, 64)
            unit["us_o"], unit["us_d"] = o.numpy(), d.numpy()
            unit["us_z0"], unit["us_sdf0"] = z.numpy(), sdf0.numpy()
            zc, sc = z, sdf0
            for i in range(4):
                znew = m32.up_sample(o, d, zc, sc, 16, 64 * 2 ** i)
                unit[f"us_znew{i}"] = znew.numpy()
                zc, sc = m32.cat_z_vals(o, d, zc, znew, sc, last=(i == 3))
                unit[f"us_zcat{i}"] = zc.numpy()
                if i < 3:
                    unit[f"us_sdfcat{i}"] = sc.numpy()

        # get_alpha (models/neus_hint_model.py:333-357)
        P = 160
        apts = (torch.rand(P, 3, generator=g) * 2 - 1) * 0.8
        adirs = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
        adists = torch.rand(P, 1, generator=g) * 0.03 + 1e-3
        for ratio in (1.0, 0.37):
            with torch.no_grad():
                a = m32.get_alpha(apts.clone(), adists, adirs, ratio)
            unit[f"alpha_r{ratio}"] = a[0].detach().numpy()
        unit["alpha_pts"], unit["alpha_dirs"], unit["alpha_dists"] = apts.numpy(), adirs.numpy(), adists.numpy()

        # reflectance network (fields/reflectance_network.py:68-96)
        with torch.no_grad():
            cp = (torch.rand(P, 3, generator=g) * 2 - 1)
            cn = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
            cv = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
            cf = torch.randn(P, 256, generator=g) * 0.3
            cl = torch.randn(P, 3, generator=g) * 3
            cvis = torch.rand(P, 1, generator=g)
            ccue = torch.rand(P, 4, generator=g) * 2
            cc = m32.color_network(cp, cn, cv, cf, cl, cvis, ccue)
        for k, v in dict(col_pts=cp, col_n=cn, col_v=cv, col_feat=cf, col_pl=cl, col_vis=cvis, col_cue=ccue,
                         col_out=cc).items():
            unit[k] = v.numpy()
        np.savez_compressed(os.path.join(HERE, f"unit_{tag}.npz"), **unit)

        # ---------------- end-to-end eval render ----------------
        N = 96
        o, d, pl, near, far = make_rays(N, seed=3, spread=0.15)
        rec = dict(o=o, d=d, pl=pl, near=near, far=far)
        for bg in (1.0, 0.0):
            for dt, model, sfx in ((torch.float32, m32, ""), (torch.float64, m64, "_f64")):
                rb = RayBundle(origins=torch.from_numpy(o).to(dt), directions=torch.from_numpy(d).to(dt),
                               pl_positions=torch.from_numpy(pl).to(dt), nears=torch.from_numpy(near).to(dt),
                               fars=torch.from_numpy(far).to(dt))
                with torch.no_grad():
                    r = model(rb, is_training=False, background_rgb=torch.full((1, 3), bg, dtype=dt))
                if bg == 0.0:
                    rec["rgb_bg0" + sfx] = r.rgb.detach().numpy()
                    continue
                for name in ("rgb", "depth", "weights", "s_val", "inside_sphere", "relax_inside_sphere",
                             "analytic_normals", "normalized_analytic_normals", "visibilities", "specular_cue"):
                    rec[name + sfx] = getattr(r, name).detach().numpy()
        np.savez_compressed(os.path.join(HERE, f"render_{tag}.npz"), **rec)

        # ---------------- training-mode forward with recorded jitter ----------------
        N = 40
        o, d, pl, near, far = make_rays(N, seed=11, spread=0.1)
        drawn = []
        real_rand = torch.rand

        def rec_rand(*a, **k):
            t = real_rand(*a, **k)
            drawn.append(t.detach().clone())
            return t

        trec = dict(o=o, d=d, pl=pl, near=near, far=far, global_step=np.int64(20000))
        torch.manual_seed(5)
        torch.rand = rec_rand
        try:
            rays_t = [torch.from_numpy(a).clone().requires_grad_(i < 3) for i, a in enumerate((o, d, pl, near, far))]
            rb = RayBundle(origins=rays_t[0], directions=rays_t[1], pl_positions=rays_t[2], nears=rays_t[3],
                           fars=rays_t[4])
            r = m32(rb, is_training=True, background_rgb=torch.ones(1, 3), global_step=20000)
        finally:
            torch.rand = real_rand
        assert len(drawn) == 2, len(drawn)
        trec["t_rand_primary"] = drawn[0].numpy()   # models/neus_hint_model.py:682
        trec["t_rand_shadow"] = drawn[1].numpy()    # models/neus_hint_model.py:394
        for name in ("rgb", "depth", "weights", "analytic_normals", "visibilities", "specular_cue", "inside_sphere"):
            trec[name] = getattr(r, name).detach().numpy()
        # loss as the caller computes it (pipelines/base_pipeline.py:57-62)
        gt = torch.from_numpy(make_rays(N, seed=99)[0][:, :3] * 0 + 0.5).float()
        rgb_loss = torch.nn.functional.l1_loss(r.rgb, gt, reduction="sum") / (N + 1e-5)
        ge = (torch.linalg.norm(r.analytic_normals, ord=2, dim=-1) - 1.0) ** 2
        eik = (r.relax_inside_sphere * ge).sum() / (r.relax_inside_sphere.sum() + 1e-5)
        loss = rgb_loss + 0.1 * eik
        trec["rgb_gt"] = gt.numpy()
        trec["loss"], trec["rgb_loss"], trec["eikonal_loss"] = (x.detach().numpy() for x in (loss, rgb_loss, eik))
        # parameter / ray gradients of that loss (trainer/trainer.py:278-279), for the backward parity tests
        m32.zero_grad()
        loss.backward()
        for name, prm in m32.named_parameters():
            trec["grad." + name] = prm.grad.detach().numpy().copy() if prm.grad is not None else np.zeros(0, np.float32)
        for nm, t in zip(("origins", "directions", "pl_positions"), rays_t):
            trec["grad.rays." + nm] = t.grad.detach().numpy().copy()
        # the same step in float64 (same drawn jitter, same ground truth): the yardstick for the gradient tolerances -
        # |grad32 - grad64| is the reference's OWN float32 noise on each tensor (tests derive their bounds from it)
        replay = [d_.double() for d_ in drawn]

        def replay_rand(*a, **k):
            return replay.pop(0)

        torch.rand = replay_rand
        try:
            rays64 = [torch.from_numpy(a).double().requires_grad_(i < 3) for i, a in enumerate((o, d, pl, near, far))]
            rb64 = RayBundle(origins=rays64[0], directions=rays64[1], pl_positions=rays64[2], nears=rays64[3], fars=rays64[4])
            r64 = m64(rb64, is_training=True, background_rgb=torch.ones(1, 3, dtype=torch.float64), global_step=20000)
        finally:
            torch.rand = real_rand
        assert not replay
        gt64 = gt.double()
        rgb_loss64 = torch.nn.functional.l1_loss(r64.rgb, gt64, reduction="sum") / (N + 1e-5)
        ge64 = (torch.linalg.norm(r64.analytic_normals, ord=2, dim=-1) - 1.0) ** 2
        eik64 = (r64.relax_inside_sphere * ge64).sum() / (r64.relax_inside_sphere.sum() + 1e-5)
        loss64 = rgb_loss64 + 0.1 * eik64
        trec["loss_f64"] = loss64.detach().numpy()
        m64.zero_grad()
        loss64.backward()
        for name, prm in m64.named_parameters():
            trec["grad64." + name] = prm.grad.detach().numpy().copy() if prm.grad is not None else np.zeros(0, np.float64)
        for nm, t in zip(("origins", "directions", "pl_positions"), rays64):
            trec["grad64.rays." + nm] = t.grad.detach().numpy().copy()
        m64.zero_grad()
        np.savez_compressed(os.path.join(HERE, f"train_{tag}.npz"), **trec)
        print("scene", tag, "done; rgb mean", float(rec["rgb"].mean()), "vis mean", float(rec["visibilities"].mean()))


    # ---------------- off-default renderer branches, scene b weights ----------------
    variants = {
        "pln": (NeuSRendererConfig(shadow_hint=False, specular_hint=False), naive_state(state_b)),      # pl-naive preset
        "ana": (NeuSRendererConfig(normal_type=NormalComputationType.Analytic), state_b),
        "mwp": (NeuSRendererConfig(depth_type=DepthComputationType.MaximalWeightPoint), state_b),
    }
    N = 64
    o, d, pl, near, far = make_rays(N, seed=23, spread=0.12)
    rec = dict(o=o, d=d, pl=pl, near=near, far=far)
    for vt, (rcfg, st) in variants.items():
        torch.manual_seed(0)
        m = NeuSHintRenderer(NeuSModelConfig(renderer=rcfg))
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
        m = m.eval()
        rb = RayBundle(origins=torch.from_numpy(o), directions=torch.from_numpy(d), pl_positions=torch.from_numpy(pl),
                       nears=torch.from_numpy(near), fars=torch.from_numpy(far))
        with torch.no_grad():
            r = m(rb, is_training=False, background_rgb=torch.ones(1, 3))
        for name in ("rgb", "depth", "weights", "visibilities", "specular_cue"):
            v = getattr(r, name)
            if v is not None:
                rec[f"{vt}.{name}"] = v.detach().numpy()
        print("variant", vt, "rgb mean", float(r.rgb.mean()), "vis is None:", r.visibilities is None)
    np.savez_compressed(os.path.join(HERE, "render_variants_b.npz"), **rec)


if __name__ == "__main__":
    main()
