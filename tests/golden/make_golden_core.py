#!/usr/bin/env python3
"""Unit fixtures for the per-ray stages of the evaluation render, recorded by IMPORTING the reference
(build container only; needs /root/reference).  Writes tests/golden/core_{a,b}.npz: the intermediates of
NeuSHintRenderer.render_core / get_visibility (models/neus_hint_model.py:475-651, :373-432) of one evaluation render -

  primary ray   z_vals -> dists, mid_z; sdf, gradients at the mid-points (get_alpha :335-336), alpha (:354), weights (:521-525),
                depths / hit_points (:531-533), hit_point_normal (:586-587), per-ray specular cue (:590-616)
  shadow ray    its get_alpha call: points' direction, dists, sdf, gradients, alpha (:417-427) and the visibility (:429-432)
  reflectance   sampled_color (:626-627) and the composite color (:635-637, white and black background)

captured with wrappers around get_alpha, F.normalize and color_network.forward (no reference code is changed or stored).
The HIP kernels behind nrh_alpha_composite / nrh_visibility / nrh_color_composite are tested against these on the GPU."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


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
    import models.neus_hint_model as ref_mod  # reference
    from models.neus_hint_model import NeuSHintRenderer, NeuSModelConfig  # reference
    from camera.ray_utils import RayBundle  # reference
    from nrhints_amd.synthetic import make_rays, perturb_state

    state_a = dict(np.load(os.path.join(HERE, "scene_a_state.npz")))
    for tag, state in (("a", state_a), ("b", perturb_state(state_a))):
        torch.manual_seed(0)
        m = NeuSHintRenderer(NeuSModelConfig())
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
        m = m.eval()
        N = 48
        o, d, pl, near, far = make_rays(N, seed=41, spread=0.14)
        rec = dict(o=o, d=d, pl=pl)
        alpha_calls, norm_calls, color_calls, core_calls = [], [], [], []
        real_get_alpha, real_core, real_color = m.get_alpha, m.render_core, m.color_network.forward
        real_normalize = ref_mod.F.normalize

        def get_alpha(pts, dists, dirs, cos_anneal_ratio=1.0):
            out = real_get_alpha(pts, dists, dirs, cos_anneal_ratio)
            with torch.no_grad():
                sdf = m.sdf_network.sdf(pts)      # the value get_alpha computed at :335 (deterministic re-evaluation)
            alpha_calls.append(dict(pts=pts.detach().clone(), dists=dists.detach().clone(), dirs=dirs.detach().clone(), sdf=sdf.detach().clone(),
                                    grad=out[2].detach().clone(), alpha=out[0].detach().clone(), inv_s=out[3].detach().clone(), ratio=cos_anneal_ratio))
            return out

        def render_core(rays_o, rays_d, rays_pl, z_vals, sample_dist, **kw):
            core_calls.append(dict(z_vals=z_vals.detach().clone(), sample_dist=float(sample_dist)))
            return real_core(rays_o, rays_d, rays_pl, z_vals, sample_dist, **kw)

        def normalize(x, *a, **k):
            y = real_normalize(x, *a, **k)
            norm_calls.append(y.detach().clone())
            return y

        def color_forward(points, normals, view_dirs, feature_vectors, pls, vis, cue):
            y = real_color(points, normals, view_dirs, feature_vectors, pls, vis, cue)
            color_calls.append(dict(out=y.detach().clone(), vis=vis.detach().clone(), cue=cue.detach().clone(), normals=normals.detach().clone()))
            return y

        m.get_alpha, m.render_core, m.color_network.forward = get_alpha, render_core, color_forward
        ref_mod.F.normalize = normalize
        try:
            outs = {}
            for bg in (1.0, 0.0):
                rb = RayBundle(origins=torch.from_numpy(o), directions=torch.from_numpy(d), pl_positions=torch.from_numpy(pl),
                               nears=torch.from_numpy(near), fars=torch.from_numpy(far))
                with torch.no_grad():
                    outs[bg] = m(rb, is_training=False, background_rgb=torch.full((1, 3), bg))
        finally:
            ref_mod.F.normalize = real_normalize
            m.get_alpha, m.render_core, m.color_network.forward = real_get_alpha, real_core, real_color
        # first render (bg = 1): get_alpha call 0 = primary mid-points, 1 = shadow ray; normalize call 0 = per-sample normals,
        # 1 = hit-point normal, 2.. = l, v, h
        prim, shad = alpha_calls[0], alpha_calls[1]
        r = outs[1.0]
        z = core_calls[0]["z_vals"]
        dists = torch.cat([z[..., 1:] - z[..., :-1], torch.tensor([core_calls[0]["sample_dist"]]).expand(N, 1)], -1)
        assert torch.equal(dists, prim["dists"])
        rec.update(mid_z=(z + dists * 0.5).numpy(), dists=dists.numpy(), sdf=prim["sdf"].reshape(N, 128).numpy(), grad=prim["grad"].numpy(),
                   alpha=prim["alpha"].reshape(N, 128).numpy(), inv_s=np.float32(prim["inv_s"][0, 0].item()),
                   weights=r.weights.numpy(), inside_sphere=r.inside_sphere.numpy(), depth=r.depth.numpy(),
                   hit_points=(torch.from_numpy(o) + torch.from_numpy(d) * r.depth).numpy(),
                   nhat=norm_calls[0].numpy(), hit_normal=norm_calls[1].numpy(), cue=r.specular_cue[:, 0, :].numpy(),
                   vis=r.visibilities.numpy())
        assert norm_calls[0].shape == (N * 128, 3) and norm_calls[1].shape == (N, 3)
        assert torch.equal(color_calls[0]["cue"].reshape(N, 128, 4)[:, 0], r.specular_cue[:, 0, :])
        # shadow ray at its 128 section mid-points
        rec.update(s_dirs=shad["dirs"].reshape(N, 128, 3)[:, 0].numpy(), s_dists=shad["dists"].numpy(), s_sdf=shad["sdf"].reshape(N, 128).numpy(),
                   s_grad=shad["grad"].numpy(), s_alpha=shad["alpha"].reshape(N, 128).numpy())
        rec.update(sampled_color=color_calls[0]["out"].reshape(N, 128, 3).numpy(), rgb=r.rgb.numpy(), rgb_bg0=outs[0.0].rgb.numpy(),
                   analytic_normals=r.analytic_normals.detach().numpy())
        np.savez_compressed(os.path.join(HERE, f"core_{tag}.npz"), **rec)
        print("scene", tag, "core fixture:", {k: v.shape for k, v in rec.items() if hasattr(v, "shape") and v.ndim > 0})


if __name__ == "__main__":
    main()
