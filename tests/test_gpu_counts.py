"""Sample counts off the reference's defaults (models/neus_hint_model.py:139-171, :696-713, :373-412): the
per-ray kernels take n_samples, n_importance_samples // up_sample_steps (at most 16 per step), n_shadow_samples and
n_shadow_importance_samples // 4 as parameters (NrhNet.n_coarse .. lin_tables); the per-sample arrays keep 128 slots per ray and
pad.  Evaluation and one training step per variant against the reference's recorded run (tests/golden/make_golden_counts.py)."""
import numpy as np
import pytest
import torch

import nrhints_amd as na
from nrhints_amd.synthetic import psnr
from tests.conftest import grad_bound, load_npz

pytestmark = pytest.mark.gpu
T = torch.from_numpy
VARIANTS = {
    "c3232": (dict(n_samples=32, n_importance_samples=32, up_sample_steps=2), 64),
    "c6432": (dict(n_samples=64, n_importance_samples=32, up_sample_steps=2), 96),
    "c4848": (dict(n_samples=48, n_importance_samples=48, up_sample_steps=4, n_shadow_samples=32, n_shadow_importance_samples=32), 96),
    "c8000": (dict(n_samples=80, n_importance_samples=0, n_shadow_samples=48, n_shadow_importance_samples=0), 80),
}


def cu(a):
    return (T(a) if isinstance(a, np.ndarray) else a).float().contiguous().cuda()


def _bundle(g, pre=""):
    return na.RayBundle(origins=cu(g[pre + "o"]), directions=cu(g[pre + "d"]), pl_positions=cu(g[pre + "pl"]), nears=cu(g[pre + "near"]),
                        fars=cu(g[pre + "far"]))


def _model(state, prec, kw):
    m = na.NeuSHintRenderer(na.NeuSModelConfig(renderer=na.NeuSRendererConfig(**kw)), precision=prec)
    m.load_state_dict({k: T(np.asarray(v)) for k, v in state.items()})
    return m.cuda()


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
@pytest.mark.parametrize("vt", sorted(VARIANTS))
def test_sample_counts_eval_vs_reference(scene_states, vt, prec):
    g = load_npz("render_counts_b.npz")
    kw, Tn = VARIANTS[vt]
    model = _model(scene_states["b"], prec, kw).eval()
    with torch.no_grad():
        out = model(_bundle(g), background_rgb=torch.ones(1, 3).cuda())
    assert out.weights.shape == (64, Tn) == g[f"{vt}.weights"].shape and out.normalized_analytic_normals.shape == (64, Tn, 3)
    rgb = out.rgb.cpu().numpy()
    assert np.abs(rgb - g[f"{vt}.rgb_f64"]).max() < 3e-5 and psnr(rgb, g[f"{vt}.rgb_f64"]) > 80.0
    np.testing.assert_allclose(out.depth.cpu().numpy(), g[f"{vt}.depth"], rtol=0, atol=3e-4)
    np.testing.assert_allclose(out.visibilities.cpu().numpy(), g[f"{vt}.visibilities"], rtol=0, atol=3e-3)
    # (the cue reaches ~2 on grazing rays; the hit normal behind it is a weighted sum over samples placed at fp32 noise: 3e-4 relative)
    np.testing.assert_allclose(out.specular_cue.cpu().numpy(), g[f"{vt}.specular_cue"], rtol=1e-3, atol=3e-4)
    w = out.weights.cpu().numpy()
    assert np.abs(w - g[f"{vt}.weights_f64"]).mean() < 2e-5 and np.abs(w.sum(1) - g[f"{vt}.weights_f64"].sum(1)).max() < 1e-4
    # (a mid-point within an ulp of the unit sphere may fall on the other side: 1 of 6 144 in the f16x3 run of 48 + 48 | 32 + 32)
    assert float((out.inside_sphere.cpu().numpy() != g[f"{vt}.inside_sphere"]).mean()) < 1e-3
    # re-chunking does not change a ray
    model.max_chunk_rays = 24
    with torch.no_grad():
        out2 = model(_bundle(g), background_rgb=torch.ones(1, 3).cuda())
    assert torch.equal(out2.rgb, out.rgb) and torch.equal(out2.weights, out.weights)


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
@pytest.mark.parametrize("vt", sorted(VARIANTS))
def test_sample_counts_training_step_vs_reference(scene_states, vt, prec):
    from nrhints_amd import train_fused
    from nrhints_amd.training import train_loss_dict
    g = load_npz("render_counts_b.npz")
    kw, Tn = VARIANTS[vt]
    model = _model(scene_states["b"], prec, kw).train()
    tb = _bundle(g, "t.")
    for t_ in (tb.origins, tb.directions, tb.pl_positions):
        t_.requires_grad_(True)
    assert train_fused.supported(model, tb) is None          # (the fused step takes them too: checked against this path below)
    out = model(tb, is_training=True, background_rgb=torch.ones(1, 3).cuda(), global_step=int(g["t.global_step"]),
                _t_rand_primary=cu(g[f"{vt}.t_rand_primary"]), _t_rand_shadow=cu(g[f"{vt}.t_rand_shadow"]))
    assert out.weights.shape == (32, Tn) and out.analytic_normals.shape == (32, Tn, 3) and out.relax_inside_sphere.shape == (32, Tn)
    np.testing.assert_allclose(out.rgb.detach().cpu().numpy(), g[f"{vt}.t.rgb"], rtol=0, atol=1e-4)
    ld = train_loss_dict(out, cu(g["t.rgb_gt"]), 0.1)
    np.testing.assert_allclose(float(ld["loss"].detach()), float(g[f"{vt}.loss"]), rtol=2e-4)
    ld["loss"].backward()
    named = dict(model.named_parameters())
    keys = [k for k in g if k.startswith(f"{vt}.grad.")]
    assert len(keys) == 14
    for k in keys:
        name = k[len(vt) + 6:]
        want64 = g[k.replace(".grad.", ".grad64.")]
        # 32 rays, one coarse draw of the reference's own noise (as for the other 32-ray fixtures: factor 4), and as few as 64
        # samples per ray: floor 1e-2 of the tensor's scale (measured: 5.1e-3 on the first reflectance layer at 32 + 32).  The
        # variance gradient is one number that cancels to 4e-6 at 80 + 0 samples (per-ray terms ~1e-4): absolute floor 3e-7
        bound, scale = grad_bound(g[k], want64, factor=4.0, floor=1e-2)
        if np.size(want64) == 1:
            bound = max(bound, 3e-7)
        got = (getattr(tb, name[5:]).grad if name.startswith("rays.") else named[name].grad).detach().cpu().numpy().astype(np.float64)
        assert got.shape == want64.shape, (vt, name)
        err = float(np.abs(got - want64).max())
        assert err <= bound, (vt, name, err, bound, scale)

    # the autograd-free step on the same batch and jitter: same kernels, same numbers to float32 round-off
    fused = _model(scene_states["b"], prec, kw).train()
    l8 = train_fused.train_step_backward(fused, _bundle(g, "t."), cu(g["t.rgb_gt"]), torch.ones(1, 3).cuda(), int(g["t.global_step"]),
                                         t_rand_primary=cu(g[f"{vt}.t_rand_primary"]), t_rand_shadow=cu(g[f"{vt}.t_rand_shadow"]))
    np.testing.assert_allclose(float(l8[0]), float(ld["loss"].detach()), rtol=5e-6)
    for (name, pa), (_, pf) in zip(model.named_parameters(), fused.named_parameters()):
        scale = float(pa.grad.abs().max()) + 1e-30
        assert float((pa.grad - pf.grad).abs().max()) < 1e-4 * scale + 5e-6, (vt, name)


HG_COUNTS = {
    "c4848g": dict(n_samples=48, n_importance_samples=48, up_sample_steps=4, n_shadow_samples=32, n_shadow_importance_samples=32),
    "c8000g": dict(n_samples=80, n_importance_samples=0, n_shadow_samples=48, n_shadow_importance_samples=0),
    "c6464g": dict(n_shadow_samples=64, n_shadow_importance_samples=32),
}


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
@pytest.mark.parametrize("vt", sorted(HG_COUNTS))
def test_shadow_hint_gradient_with_sample_counts_vs_reference(scene_states, vt, prec):
    """renderer.shadow_hint_gradient with shadow-ray counts off the defaults (models/neus_hint_model.py:379, :411-432):
    the differentiable visibility (ShadowVisibilityHip -> nrh_shadow_alpha_forward / _backward, n_real = the shadow ray's
    existing samples) is the transmittance in front of the LAST EXISTING sample - before ABI 147 the product ran to slot 127,
    through the last real sample and the padded slots (alpha = 1e-5 / (cdf + 1e-5) there, not 0).  One training step against
    the reference's recorded run (tests/golden/make_golden_counts_hintgrad.py): rgb, visibilities, loss, 11 gradient tensors -
    d loss / d variance exists only through the visibility's gradient path here."""
    from nrhints_amd.training import train_loss_dict
    g = load_npz("render_counts_hintgrad_b.npz")
    model = _model(scene_states["b"], prec, dict(shadow_hint_gradient=True, **HG_COUNTS[vt])).train()
    tb = _bundle(g, "t.")
    out = model(tb, is_training=True, background_rgb=torch.ones(1, 3).cuda(), global_step=int(g["t.global_step"]),
                _t_rand_primary=cu(g[f"{vt}.t_rand_primary"]), _t_rand_shadow=cu(g[f"{vt}.t_rand_shadow"]))
    np.testing.assert_allclose(out.rgb.detach().cpu().numpy(), g[f"{vt}.t.rgb"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(out.visibilities.detach().cpu().numpy(), g[f"{vt}.t.visibilities"], rtol=0, atol=3e-3)
    ld = train_loss_dict(out, cu(g["t.rgb_gt"]), 0.1)
    np.testing.assert_allclose(float(ld["loss"].detach()), float(g[f"{vt}.loss"]), rtol=2e-4)
    ld["loss"].backward()
    named = dict(model.named_parameters())
    keys = [k for k in g if k.startswith(f"{vt}.grad.")]
    assert len(keys) == 11
    for k in keys:
        name = k[len(vt) + 6:]
        want64 = g[k.replace(".grad.", ".grad64.")]
        bound, scale = grad_bound(g[k], want64, factor=4.0, floor=1e-2)
        if np.size(want64) == 1:
            bound = max(bound, 3e-7)
        err = float(np.abs(named[name].grad.detach().cpu().numpy().astype(np.float64) - want64).max())
        assert err <= bound, (vt, name, err, bound, scale)
    # ... and the fused (autograd-free) step on the same batch: the same kernels, n_real handed to the shadow alpha stage there too
    from nrhints_amd import train_fused
    fused = _model(scene_states["b"], prec, dict(shadow_hint_gradient=True, **HG_COUNTS[vt])).train()
    assert train_fused.supported(fused, _bundle(g, "t.")) is None
    l8 = train_fused.train_step_backward(fused, _bundle(g, "t."), cu(g["t.rgb_gt"]), torch.ones(1, 3).cuda(), int(g["t.global_step"]),
                                         t_rand_primary=cu(g[f"{vt}.t_rand_primary"]), t_rand_shadow=cu(g[f"{vt}.t_rand_shadow"]))
    np.testing.assert_allclose(float(l8[0]), float(ld["loss"].detach()), rtol=5e-6)
    for (name, pa), (_, pf) in zip(model.named_parameters(), fused.named_parameters()):
        scale = float(pa.grad.abs().max()) + 1e-30
        assert float((pa.grad - pf.grad).abs().max()) < 3e-4 * scale + 5e-6, (vt, name)
