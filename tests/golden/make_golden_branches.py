#!/usr/bin/env python3
"""Golden fixture for further off-default renderer branches, recorded by IMPORTING the
reference.  Build container only; writes tests/golden/render_branches_b.npz (plain data).

    python tests/golden/make_golden_branches.py

Variants, all on scene b's weights (nrhints_amd.synthetic.perturb_state of the reference init):
  st    depth_type = SphereTracing (models/neus_hint_model.py:359-372, :527-528): the traced points / depths on their own and
        the whole evaluation render that feeds its hints from them
  sho   shadow hint only   (shadow_hint=True,  specular_hint=False): evaluation render + one training step's loss and gradients
  spo   specular hint only (shadow_hint=False, specular_hint=True):  the same
  frc   force_shadow_map + force_specular_cue on top of both hints (a no-op: has_*_hint = hint or force, :239-240)
  i0    n_importance_samples = 0 with both hints off (BASELINE configs[0]'s plumbing variant, SURVEY 8d C1: 64 samples per ray,
        :696): evaluation render + one training step
  psh   n_shadow_importance_clip = 8 (:553-575: a shadow ray per group of 16 samples): evaluation render + one training step
  shg / spg / bhg   shadow_hint_gradient / specular_hint_gradient / both (:379, :589: the hints stay inside the autograd graph):
        one training step's loss and gradients (the forward values equal the default model's)
and the reference's own failure for force_* WITHOUT the hint (recorded as the exception's type name).
"""
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


# gradient tensors kept per training step (all of them would be 9 MB per variant): the first reflectance layer (whose shape the
# variant changes) and one tensor of every kind elsewhere
KEEP_GRADS = ("color_network.lin0.weight_v", "color_network.lin0.weight_g", "color_network.lin0.bias", "color_network.lin4.weight_v",
              "color_network.lin2.bias", "sdf_network.lin0.weight_v", "sdf_network.lin4.weight_g", "sdf_network.lin7.bias",
              "sdf_network.out_sdf.weight_v", "sdf_network.out_feat.weight_g", "deviation_network.variance")


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)
    import torch

    torch.set_num_threads(8)
    from camera.ray_utils import RayBundle
    from models.neus_hint_model import DepthComputationType, NeuSHintRenderer, NeuSModelConfig, NeuSRendererConfig

    from nrhints_amd.synthetic import make_rays, naive_state, one_hint_state, perturb_state

    state_b = perturb_state(dict(np.load(os.path.join(HERE, "scene_a_state.npz"))))

    def build(rcfg, st, dtype=torch.float32):
        torch.manual_seed(0)
        m = NeuSHintRenderer(NeuSModelConfig(renderer=rcfg))
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
        return m.to(dtype)

    def bundle(arrs, dtype=torch.float32, grad=False):
        ts = [torch.from_numpy(a).to(dtype).clone().requires_grad_(grad and i < 3) for i, a in enumerate(arrs)]
        return RayBundle(origins=ts[0], directions=ts[1], pl_positions=ts[2], nears=ts[3], fars=ts[4]), ts

    R = NeuSRendererConfig
    variants = {
        "st": (R(depth_type=DepthComputationType.SphereTracing), state_b),
        "sho": (R(shadow_hint=True, specular_hint=False), one_hint_state(state_b, shadow=True)),
        "spo": (R(shadow_hint=False, specular_hint=True), one_hint_state(state_b, shadow=False)),
        "frc": (R(force_shadow_map=True, force_specular_cue=True), state_b),
        "psh": (R(n_shadow_importance_clip=8), state_b),
        "i0": (R(n_importance_samples=0, shadow_hint=False, specular_hint=False), naive_state(state_b)),
    }
    grad_variants = {
        "shg": (R(shadow_hint_gradient=True), state_b),
        "spg": (R(specular_hint_gradient=True), state_b),
        "bhg": (R(shadow_hint_gradient=True, specular_hint_gradient=True), state_b),
    }
    N = 64
    rays = make_rays(N, seed=29, spread=0.12)
    rec = dict(zip(("o", "d", "pl", "near", "far"), rays))
    for vt, (rcfg, st) in variants.items():
        m = build(rcfg, st).eval()
        rb, _ = bundle(rays)
        with torch.no_grad():
            r = m(rb, is_training=False, background_rgb=torch.ones(1, 3))
            if vt == "st":
                pts, dep = m.sphere_trace(rb.origins, rb.directions, 2000, 1e-4, 100)
                rec["st.trace_pts"], rec["st.trace_depths"] = pts.numpy(), dep.numpy()
                pts64, dep64 = build(rcfg, st, torch.float64).sphere_trace(rb.origins.double(), rb.directions.double(), 2000, 1e-4, 100)
                rec["st.trace_pts_f64"], rec["st.trace_depths_f64"] = pts64.numpy(), dep64.numpy()
        for name in ("rgb", "depth", "weights", "visibilities", "specular_cue") + (("inside_sphere", "normalized_analytic_normals", "s_val") if vt == "i0" else ()):
            v = getattr(r, name)
            rec[f"{vt}.{name}"] = v.detach().numpy() if v is not None else np.zeros(0, np.float32)
        print("variant", vt, "rgb mean", float(r.rgb.mean()), "depth mean", float(r.depth.mean()))

    # one training step for the one-hint models, jitter recorded (as make_golden.py does for the full model), fp32 and fp64
    Nt = 32
    trays = make_rays(Nt, seed=31, spread=0.1)
    rec.update({"t." + k: v for k, v in zip(("o", "d", "pl", "near", "far"), trays)})
    gt = torch.full((Nt, 3), 0.5)
    rec["t.rgb_gt"], rec["t.global_step"] = gt.numpy(), np.int64(20000)
    real_rand = torch.rand
    for vt in ("sho", "spo", "shg", "spg", "bhg", "psh", "i0"):
        rcfg, st = variants[vt] if vt in variants else grad_variants[vt]
        drawn = []

        def rec_rand(*a, **k):
            t = real_rand(*a, **k)
            drawn.append(t.detach().clone())
            return t

        for dt, sfx in ((torch.float32, ""), (torch.float64, "64")):
            m = build(rcfg, st, dt).train()
            replay = [x.to(dt) for x in drawn]
            torch.manual_seed(5)
            torch.rand = rec_rand if dt == torch.float32 else (lambda *a, **k: replay.pop(0))
            try:
                rb, ts = bundle(trays, dt, grad=True)
                r = m(rb, is_training=True, background_rgb=torch.ones(1, 3, dtype=dt), global_step=20000)
            finally:
                torch.rand = real_rand
            g = gt.to(dt)
            rgb_loss = torch.nn.functional.l1_loss(r.rgb, g, reduction="sum") / (Nt + 1e-5)
            ge = (torch.linalg.norm(r.analytic_normals, ord=2, dim=-1) - 1.0) ** 2
            eik = (r.relax_inside_sphere * ge).sum() / (r.relax_inside_sphere.sum() + 1e-5)
            loss = rgb_loss + 0.1 * eik
            loss.backward()
            if dt == torch.float32:
                # shadow-only draws primary + shadow jitter; specular-only has no shadow march, hence one draw
                assert len(drawn) == (1 if vt in ("spo", "i0") else 2), len(drawn)
                rec[f"{vt}.t_rand_primary"] = drawn[0].numpy()
                if vt not in ("spo", "i0"):
                    rec[f"{vt}.t_rand_shadow"] = drawn[1].numpy()
                rec[f"{vt}.t.rgb"] = r.rgb.detach().numpy()
            rec[f"{vt}.loss{sfx}"] = loss.detach().numpy()
            for name, prm in m.named_parameters():
                if name in KEEP_GRADS:
                    rec[f"{vt}.grad{sfx}.{name}"] = prm.grad.detach().numpy().copy()
            for nm, t in zip(("origins", "directions", "pl_positions"), ts):
                rec[f"{vt}.grad{sfx}.rays.{nm}"] = t.grad.detach().numpy().copy()

    # force_* without the hint: what the reference does
    for vt, rcfg, st in (("force_shadow_only", R(shadow_hint=False, specular_hint=False, force_shadow_map=True), None),
                         ("force_specular_only", R(shadow_hint=False, specular_hint=False, force_specular_cue=True), None)):
        torch.manual_seed(0)
        m = NeuSHintRenderer(NeuSModelConfig(renderer=rcfg)).eval()
        rb, _ = bundle(tuple(a[:4] for a in rays))
        try:
            with torch.no_grad():
                m(rb, is_training=False, background_rgb=torch.ones(1, 3))
            outcome = "ok"
        except Exception as e:  # noqa: BLE001 - recording whatever the reference raises
            outcome = type(e).__name__ + ": " + str(e).splitlines()[0][:120]
        rec[vt + ".outcome"] = np.array(outcome)
        print(vt, "->", outcome)
    np.savez_compressed(os.path.join(HERE, "render_branches_b.npz"), **rec)


if __name__ == "__main__":
    main()
