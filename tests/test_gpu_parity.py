"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the committed golden fixtures.

Tolerances are float32 ones and are written next to each assertion.  The reference's own fp32-vs-fp64 noise
floor on this path: rgb <= 1e-5, depth 6e-5, visibilities 2.5e-4, per-sample weights up to
3.7e-3 at isolated samples with mean 1e-6 - hence mean + outlier budgets for per-sample fields.
"""
import os

import numpy as np
import pytest
import torch

import nrhints_amd as na
from nrhints_amd import ops, packing as pk
from nrhints_amd.synthetic import make_rays, psnr
from oracle import neus_oracle as orc
from tests.conftest import load_npz

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def cu(a):
    return (T(a) if isinstance(a, np.ndarray) else a).float().contiguous().cuda()


@pytest.fixture(scope="module", params=["a-f32", "b-f32", "a-f16x3", "b-f16x3"])
def scene(request, scene_states):
    """scene (a: reference init, b: perturbed, sharp) x matrix arithmetic (exact fp32 MFMA | fp16 3-term split)."""
    tag, prec = request.param.split("-")
    st = scene_states[tag]
    model = na.NeuSHintRenderer(na.NeuSModelConfig(), precision=prec)
    model.load_state_dict({k: T(np.asarray(v)) for k, v in st.items()})
    model = model.cuda().eval()
    packed = model.packed_params(torch.device("cuda", torch.cuda.current_device()))
    return tag, model, packed, orc.params_from_state(st), orc.params_from_state(st, torch.float64)


def test_native_library_loaded():
    from nrhints_amd import _lib
    lib = _lib.load()           # (raises StaleLibrary if the binary was built from other sources than the tree's)
    assert lib.nrh_version() == 148
    # the binary that runs IS the tree: the hash the Makefile embedded against the hash of the sources beside it
    ident = _lib.library_identity()
    assert ident["embedded"] == ident["tree"] != "unknown", ident
    assert ident["embedded"] in ident["build_info"]
    assert lib.nrh_mlp_grid() > 0
    assert _lib.param_sizes()[:5] == [pk.SDF_PACKED_FLOATS, pk.SDF_BIAS_FLOATS, pk.SDF_HEAD_FLOATS,
                                      pk.COL_PACKED_FLOATS, pk.COL_BIAS_FLOATS]


@pytest.mark.parametrize("npts", [1, 37, 64, 1000, 4096 + 5])
def test_sdf_modes_vs_oracle(scene, npts):
    tag, model, packed, p32, p64 = scene
    g = torch.Generator().manual_seed(npts)
    pts = (torch.rand(npts, 3, generator=g) * 2 - 1) * 0.95
    o_sdf, o_feat, o_grad = orc.sdf_forward_grad_analytic(p64, pts.double())
    for mode in (0, 1, 2):
        sdf, grad, feat = ops.sdf_at_points(mode, packed["sdf_w"], packed["sdf_b"], packed["sdf_head"], pts.cuda())
        # fp32 MFMA chain vs fp64 oracle: |sdf| ~ 1, 8 layers of K=256 fp32 accumulation
        np.testing.assert_allclose(sdf.cpu().numpy()[:, 0], o_sdf.numpy()[:, 0], rtol=0, atol=5e-6)
        if mode >= 1:
            # gradient magnitude ~1 (scene b up to ~3); fp32 reverse chain
            np.testing.assert_allclose(grad.cpu().numpy(), o_grad.numpy(), rtol=0, atol=1e-4 if tag == "a" else 5e-4)
        if mode == 2:
            f = pk.feat_tiles_to_rows(feat.cpu(), npts).numpy()
            np.testing.assert_allclose(f, o_feat.numpy(), rtol=0, atol=3e-5)


def test_sdf_golden_fixture(scene):
    """Directly against what the imported reference produced (tests/golden/unit_*.npz)."""
    tag, model, packed, _, _ = scene
    u = load_npz(f"unit_{tag}.npz")
    sdf, grad, feat = ops.sdf_at_points(2, packed["sdf_w"], packed["sdf_b"], packed["sdf_head"], cu(u["sdf_pts"]))
    P = u["sdf_pts"].shape[0]
    out = np.concatenate([sdf.cpu().numpy(), pk.feat_tiles_to_rows(feat.cpu(), P).numpy()], axis=1)
    np.testing.assert_allclose(out, u["sdf_out_f64"], rtol=0, atol=3e-5)
    np.testing.assert_allclose(out, u["sdf_out"], rtol=0, atol=3e-5)
    np.testing.assert_allclose(grad.cpu().numpy(), u["sdf_grad_f64"], rtol=0, atol=1e-4 if tag == "a" else 5e-4)


def test_sdf_along_rays_strided(scene):
    """Ray-parametrised points with a row stride (the sampler's calling convention)."""
    tag, model, packed, p32, p64 = scene
    o, d, pl, near, far = make_rays(50, seed=5, spread=0.1)
    z = np.zeros((50, 128), np.float32)
    z[:, :24] = near + (far - near) * np.linspace(0, 1, 24, dtype=np.float32)[None]
    sdf, _, _ = ops.sdf_eval(0, packed["sdf_w"], packed["sdf_b"], packed["sdf_head"], cu(o), cu(d), cu(z), 24, t_stride=128)
    pts = (T(o)[:, None] + T(d)[:, None] * T(z[:, :24])[..., None]).reshape(-1, 3)
    ref = orc.sdf_forward(p64, pts.double(), False)[0].reshape(50, 24)
    np.testing.assert_allclose(sdf.cpu().numpy(), ref.numpy(), rtol=0, atol=5e-6)


def test_sampler_steps_vs_golden(scene):
    tag, model, packed, p32, _ = scene
    u = load_npz(f"unit_{tag}.npz")
    o, d = cu(u["us_o"]), cu(u["us_d"])
    N = o.shape[0]
    lin16 = torch.linspace(0, 1, 16).cuda()
    z = torch.zeros(N, 128, device="cuda")
    s = torch.zeros(N, 128, device="cuda")
    z[:, :64], s[:, :64] = cu(u["us_z0"]), cu(u["us_sdf0"])
    n = 64
    for i in range(4):
        znew, _, _ = ops.sampler_step(o, d, z, s, n, upsample_inv_s=64.0 * 2 ** i, lin16=lin16)
        got, want = znew.cpu().numpy(), u[f"us_znew{i}"]
        # Inverse-CDF sampling is ill-conditioned where the pdf sits at its 1e-5 floor (dz/du ~ bin / 1e-5): the
        # reference's OWN fp32 result differs from its fp64 result by > 2e-5 on 6 % of these samples, with a
        # maximum of one full bin (0.03) - measured on this fixture.  Budget: well inside that noise floor.
        diff = np.abs(got - want)
        assert (diff > 2e-5).mean() < 3e-2, (i, (diff > 2e-5).sum(), diff.max())
        assert np.median(diff) < 1e-6 and diff.max() < 3.2e-2, (i, np.median(diff), diff.max())
        # continue from the recorded samples so one flipped bin cannot cascade
        znew_ref = cu(want)
        if i < 3:
            pts = (o[:, None] + d[:, None] * znew_ref[..., None]).reshape(-1, 3)
            snew = orc.sdf_forward(p32, pts.cpu(), False)[0].reshape(N, 16).cuda().contiguous()
            ops.sampler_step(o, d, z, s, n, znew_in=znew_ref, snew_in=snew)
            n += 16
            np.testing.assert_array_equal(z[:, :n].cpu().numpy(), u[f"us_zcat{i}"])
            np.testing.assert_allclose(s[:, :n].cpu().numpy(), u[f"us_sdfcat{i}"], rtol=0, atol=3e-6)
            s[:, :n] = cu(u[f"us_sdfcat{i}"])
        else:
            _, tmid, dists = ops.sampler_step(o, d, z, s, n, znew_in=znew_ref, finalize=True)
            n += 16
            zc = u[f"us_zcat{i}"]
            np.testing.assert_array_equal(z.cpu().numpy(), zc)
            dd = np.concatenate([zc[:, 1:] - zc[:, :-1], np.full((N, 1), 2.0 / 64, np.float32)], axis=1)
            np.testing.assert_array_equal(dists.cpu().numpy(), dd)
            np.testing.assert_array_equal(tmid.cpu().numpy(), zc + dd * np.float32(0.5))


def test_color_vs_golden(scene):
    tag, model, packed, p32, _ = scene
    u = load_npz(f"unit_{tag}.npz")
    # the fixture has 160 free points; lay them out as 2 "rays" of 128 samples (pad by repetition) with
    # ro = 0, rd = 0 and the point carried by ... -> instead rebuild per-ray inputs: use points on real rays
    N = 3
    o, d, pl, near, far = make_rays(N, seed=21, spread=0.1)
    g = torch.Generator().manual_seed(3)
    tmid = torch.rand(N, 128, generator=g) * 2 + 2
    nhat = torch.nn.functional.normalize(torch.randn(N * 128, 3, generator=g), dim=-1)
    feat = torch.randn(N * 128, 256, generator=g) * 0.3
    vis = torch.rand(N, 1, generator=g)
    cue = torch.rand(N, 4, generator=g) * 2
    raymisc = torch.zeros(N, pk.RAYMISC_STRIDE)
    raymisc[:, 0:27] = orc.nerf_encode(T(d), 4)
    raymisc[:, 27:54] = orc.nerf_encode(T(pl), 4)
    raymisc[:, 54:63] = orc.nerf_encode(vis, 4)
    raymisc[:, 63:99] = orc.nerf_encode(cue, 4)
    col = ops.color_eval(packed["col_w"], packed["col_b"], pk.rows_to_feat_tiles(feat).cuda(), cu(o), cu(d),
                         tmid.cuda(), nhat.cuda().contiguous(), raymisc.cuda())
    pts = (T(o)[:, None] + T(d)[:, None] * tmid[..., None]).reshape(-1, 3)
    rep = lambda x: x[:, None, :].expand(N, 128, x.shape[-1]).reshape(N * 128, -1)
    ref = orc.color_forward(p32, pts, nhat, rep(T(d)), feat, rep(T(pl)), rep(vis), rep(cue))
    np.testing.assert_allclose(col.cpu().numpy(), ref.numpy(), rtol=0, atol=5e-6)  # sigmoid output in [0,1]


FIELDS_PER_SAMPLE = (("weights", 3e-5, 3e-2), ("analytic_normals", 3e-4, 0.3),
                     ("normalized_analytic_normals", 3e-4, 0.6), ("specular_cue", 2e-3, 0.1))


def _bundle(o, d, pl, near, far):
    return na.RayBundle(origins=cu(o), directions=cu(d), pl_positions=cu(pl), nears=cu(near), fars=cu(far))


def _check_against(out, g, sfx=""):
    rgb = out.rgb.cpu().numpy()
    # headline tolerance (north_star): rgb within 1e-4 absolute and PSNR(ours, reference) >= 80 dB.  The assertions sit at
    # ~3x the reference's own fp32-vs-fp64 distance on these fixtures (rgb 1e-5, depth 6e-5, visibilities 2.5e-4)
    np.testing.assert_allclose(rgb, g["rgb" + sfx], rtol=0, atol=3e-5)
    assert psnr(rgb, g["rgb" + sfx]) > 80.0
    np.testing.assert_allclose(out.depth.cpu().numpy(), g["depth" + sfx], rtol=0, atol=2e-4)
    np.testing.assert_allclose(out.visibilities.cpu().numpy(), g["visibilities" + sfx], rtol=0, atol=8e-4)
    assert np.mean(out.inside_sphere.cpu().numpy() != g["inside_sphere" + sfx]) < 2e-3
    for k, mean_tol, max_tol in FIELDS_PER_SAMPLE:
        diff = np.abs(getattr(out, k).cpu().numpy() - g[k + sfx])
        # the yardstick for per-sample fields is the reference's own fp32-vs-fp64 distance on this fixture (scene b:
        # normals differ by 1e-3 on average and 1.1 at isolated samples, because individual sample positions move)
        noise = np.abs(g[k].astype(np.float64) - g[k + "_f64"])
        assert diff.mean() < max(mean_tol, 2.0 * noise.mean()), (k, diff.mean(), noise.mean())
        assert diff.max() < max(max_tol, 1.5 * noise.max()), (k, diff.max(), noise.max())
        # isolated samples only: no more outliers than a small multiple of what the reference shows against itself
        assert (diff > max_tol).mean() < max(1e-3, 3.0 * (noise > max_tol).mean()), (k, (diff > max_tol).sum())


def test_render_eval_vs_golden(scene):
    tag, model, packed, p32, _ = scene
    g = load_npz(f"render_{tag}.npz")
    rb = _bundle(*(g[k] for k in ("o", "d", "pl", "near", "far")))
    with torch.no_grad():
        out = model(rb, is_training=False, background_rgb=torch.ones(1, 3).cuda())
        out0 = model(rb, is_training=False, background_rgb=torch.zeros(1, 3).cuda())
    assert out.rgb.shape == (96, 3) and out.weights.shape == (96, 128) and out.specular_cue.shape == (96, 128, 4)
    assert out.relax_inside_sphere.data_ptr() == out.inside_sphere.data_ptr()  # upstream quirk kept (:745)
    _check_against(out, g)            # vs reference fp32
    _check_against(out, g, "_f64")    # vs reference fp64
    np.testing.assert_allclose(out0.rgb.cpu().numpy(), g["rgb_bg0"], rtol=0, atol=1e-4)
    inv_s = float(np.exp(10.0 * (0.3 if tag == "a" else 0.7)))
    np.testing.assert_allclose(out.s_val.cpu().numpy(), g["s_val"], rtol=1e-5)
    assert abs(1.0 / out.s_val[0, 0].item() - inv_s) / inv_s < 1e-5


def test_render_training_values_vs_golden(scene):
    tag, model, packed, p32, _ = scene
    g = load_npz(f"train_{tag}.npz")
    rb = _bundle(*(g[k] for k in ("o", "d", "pl", "near", "far")))
    with torch.no_grad():
        out = model(rb, is_training=True, background_rgb=torch.ones(1, 3).cuda(), global_step=int(g["global_step"]),
                    _t_rand_primary=cu(g["t_rand_primary"]), _t_rand_shadow=cu(g["t_rand_shadow"]))
    np.testing.assert_allclose(out.rgb.cpu().numpy(), g["rgb"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(out.visibilities.cpu().numpy(), g["visibilities"], rtol=0, atol=3e-3)
    np.testing.assert_allclose(out.depth.cpu().numpy(), g["depth"], rtol=0, atol=3e-4)
    o = {k: getattr(out, k).cpu() for k in ("rgb", "analytic_normals", "relax_inside_sphere")}
    loss, rgb_loss, eik = orc.train_loss(o, T(g["rgb_gt"]))
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=2e-4)


def test_render_vs_oracle_many_rays(scene):
    """1 000 rays incl. misses; oracle in 'minimal' mode on the CPU (a few seconds)."""
    tag, model, packed, p32, _ = scene
    rays = make_rays(1000, seed=17, spread=0.15)
    with torch.no_grad():
        out = model(_bundle(*rays), background_rgb=torch.ones(1, 3).cuda())
    ref = orc.render_chunked(p32, *(T(a) for a in rays), chunk=500, background_rgb=torch.ones(1, 3), mode="minimal")
    rgb = out.rgb.cpu().numpy()
    assert psnr(rgb, ref["rgb"].numpy()) > 80.0
    d = np.abs(rgb - ref["rgb"].numpy())
    assert d.max() < 5e-4 and d.mean() < 5e-6, (d.max(), d.mean())
    dv = np.abs(out.visibilities.cpu().numpy() - ref["visibilities"].numpy())
    assert dv.mean() < 1e-4 and dv.max() < 2e-2


def test_render_properties_full_size(scene):
    """Size-independent invariants at a BASELINE-sized batch (config 1: 4096 rays, then 40 000)."""
    tag, model, packed, _, _ = scene
    for n in (4096, 40000):
        rays = make_rays(n, seed=n, spread=0.12)
        rb = _bundle(*rays)
        with torch.no_grad():
            a = model(rb, background_rgb=torch.ones(1, 3).cuda())
            b = model(rb, background_rgb=torch.zeros(1, 3).cuda())
        assert torch.isfinite(a.rgb).all() and torch.isfinite(a.weights).all()
        wsum = a.weights.sum(-1, keepdim=True)
        assert (a.weights >= 0).all() and (wsum <= 1.0 + 1e-4).all()
        assert (a.rgb >= -1e-6).all() and (a.rgb <= 1.0 + 1e-5).all()
        assert (a.visibilities >= 0).all() and (a.visibilities <= 1.0 + 1e-6).all()
        # composite is affine in the background: rgb(bg=1) - rgb(bg=0) = 1 - sum(w)
        torch.testing.assert_close(a.rgb - b.rgb, (1.0 - wsum).expand(-1, 3), rtol=0, atol=2e-6)
        # everything except rgb is background independent, and the render is deterministic
        assert torch.equal(a.weights, b.weights) and torch.equal(a.depth, b.depth)
        # unit normals where the gradient is non-degenerate
        nn_ = a.normalized_analytic_normals.norm(dim=-1)
        assert ((nn_ - 1).abs() < 1e-4).all()
        # rays that miss the unit sphere by a margin see (almost) nothing
        o, d = T(rays[0]), T(rays[1])
        closest = torch.linalg.norm(o - (o * d).sum(-1, keepdim=True) * d, dim=-1)
        miss = (closest > 1.05).cuda()
        if miss.any():
            assert wsum[miss].max() < 0.05
        # chunk independence: the same rays in differently sized C calls give bit-identical pixels
        model.max_chunk_rays = 1536
        with torch.no_grad():
            c = model(rb, background_rgb=torch.ones(1, 3).cuda())
        model.max_chunk_rays = type(model).max_chunk_rays
        assert torch.equal(a.rgb, c.rgb) and torch.equal(a.visibilities, c.visibilities)


def test_edge_cases(scene):
    tag, model, packed, _, _ = scene
    rays = make_rays(1, seed=1)
    with torch.no_grad():
        one = model(_bundle(*rays), background_rgb=torch.ones(1, 3).cuda())
    assert one.rgb.shape == (1, 3) and torch.isfinite(one.rgb).all()
    empty = [np.zeros((0, 3), np.float32)] * 3 + [np.zeros((0, 1), np.float32)] * 2
    with torch.no_grad():
        e = model(_bundle(*empty), background_rgb=torch.ones(1, 3).cuda())
    assert e.rgb.shape == (0, 3) and e.weights.shape == (0, 128)
    # CPU tensors are refused (no silent fallback)
    with pytest.raises(RuntimeError):
        model(na.RayBundle(*(T(a) for a in rays[:3]), nears=T(rays[3]), fars=T(rays[4])))


def test_sdf_query_and_grid(scene):
    tag, model, packed, p32, _ = scene
    pts = (torch.rand(300, 3) * 2 - 1)
    got = model.sdf(pts.cuda()).cpu()
    ref = orc.sdf_forward(p32, pts, False)[0]
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=0, atol=5e-6)
    u = model.extract_fields([-1, -1, -1], [1, 1, 1], 24)
    assert u.shape == (24, 24, 24) and np.isfinite(u).all()
    assert u[12, 12, 12] > 0 > u[0, 0, 0]  # -sdf: positive inside, negative outside


@pytest.mark.parametrize("sdf_backward", ["hip", "manual", "autograd"])
def test_training_step_gradients(scene, sdf_backward):
    """forward(is_training=True) + the reference's loss + backward: loss value and the gradient of every parameter
    tensor and of the rays against what the imported reference produced (tests/golden/train_*.npz)."""
    tag, model, packed, p32, _ = scene
    g = load_npz(f"train_{tag}.npz")
    rb = _bundle(*(g[k] for k in ("o", "d", "pl", "near", "far")))
    for t_ in (rb.origins, rb.directions, rb.pl_positions):
        t_.requires_grad_(True)
    model.zero_grad()
    # HIP sweeps (the product) | the same maths in torch ops | second-order autograd like the reference (tests/torch_backends.py)
    import contextlib
    from tests.torch_backends import use_torch_backend
    with (contextlib.nullcontext() if sdf_backward == "hip" else use_torch_backend(model, sdf_backward)):
        out = model(rb, is_training=True, background_rgb=torch.ones(1, 3).cuda(), global_step=int(g["global_step"]),
                    _t_rand_primary=cu(g["t_rand_primary"]), _t_rand_shadow=cu(g["t_rand_shadow"]))
    gt = cu(g["rgb_gt"])
    rgb_loss = (out.rgb - gt).abs().sum() / (out.rgb.shape[0] + 1e-5)            # pipelines/base_pipeline.py:57-62
    ge = (torch.linalg.norm(out.analytic_normals, dim=-1) - 1.0) ** 2
    eik = (out.relax_inside_sphere * ge).sum() / (out.relax_inside_sphere.sum() + 1e-5)
    loss = rgb_loss + 0.1 * eik
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=2e-4)
    # every tensor against the reference's float64 gradient, bounded by the reference's OWN float32-vs-float64 distance on
    # that tensor (tests/conftest.py grad_bound: 3 x |ref32 - ref64|, floor 1e-4 of the scale) instead of a blanket 2 %
    from tests.conftest import grad_bound
    report = []
    for name, prm in model.named_parameters():
        tol, scale = grad_bound(g["grad." + name], g["grad64." + name])
        err = float(np.abs(prm.grad.detach().cpu().numpy().astype(np.float64) - g["grad64." + name]).max())
        report.append((err / tol, name, err / scale, tol / scale))
    for nm, t_ in (("origins", rb.origins), ("directions", rb.directions), ("pl_positions", rb.pl_positions)):
        tol, scale = grad_bound(g["grad.rays." + nm], g["grad64.rays." + nm])
        err = float(np.abs(t_.grad.cpu().numpy().astype(np.float64) - g["grad64.rays." + nm]).max())
        report.append((err / tol, "rays." + nm, err / scale, tol / scale))
    bad = [r for r in report if r[0] >= 1.0]
    assert not bad, "gradient outside its derived bound (ratio, tensor, err/scale, bound/scale): " + repr(sorted(bad, reverse=True)[:8])
    model.zero_grad()


def test_eval_image_products_and_raygen(scene):
    """Fused ray generation + on-device image reductions against the reference's formulas."""
    from nrhints_amd.pipeline import CameraModel, generate_rays, render_image
    from nrhints_amd.synthetic import make_image_rays
    tag, model, packed, p32, _ = scene
    cam = CameraModel(H=24, W=32, cx=16.0, cy=12.0, fx=40.0, fy=40.0)
    # pose / light of the synthetic orbit camera
    o_np, d_np, pl_np, near_np, far_np = make_image_rays(24, 32, focal=40.0)
    # rebuild the pose make_image_rays used: origin + the three direction vectors of pixel rays are enough to check
    # generate_rays against it through an explicit pose
    import numpy as _np
    ce, se, ca, sa = _np.cos(0.5), _np.sin(0.5), _np.cos(0.6), _np.sin(0.6)
    pos = 4.0 * _np.array([ce * ca, ce * sa, se])
    fwd = -pos / _np.linalg.norm(pos)
    right = _np.cross(fwd, _np.array([0.0, 0.0, 1.0])); right /= _np.linalg.norm(right)
    up = _np.cross(right, fwd)
    pose = torch.tensor(_np.concatenate([_np.stack([right, up, -fwd], axis=1), pos[:, None]], axis=1), dtype=torch.float32)
    rb = generate_rays(cam, pose, T(pl_np[0]), "cuda")
    np.testing.assert_allclose(rb.directions.cpu().numpy(), d_np, rtol=0, atol=2e-6)
    np.testing.assert_allclose(rb.origins.cpu().numpy(), o_np, rtol=0, atol=1e-6)
    np.testing.assert_allclose(rb.nears.cpu().numpy(), near_np, rtol=0, atol=2e-5)
    np.testing.assert_allclose(rb.fars.cpu().numpy(), far_np, rtol=0, atol=2e-5)
    # image products == reductions of the full RenderOutput (pipelines/base_pipeline.py:125-131)
    img = render_image(model, cam, pose, T(pl_np[0]), white_background=True)
    with torch.no_grad():
        full = model(rb, background_rgb=torch.ones(1, 3).cuda())
    assert torch.equal(img["rgb"].reshape(-1, 3), full.rgb)
    rot = torch.linalg.inv(pose[:3, :3]).cuda()
    for key, field in (("analytic_normals", full.analytic_normals), ("normalized_analytic_normals", full.normalized_analytic_normals)):
        ref = torch.einsum("...ij,...i,...i->...j", field, full.weights, full.inside_sphere)
        ref = (rot[None] @ ref[:, :, None])[:, :, 0]
        torch.testing.assert_close(img[key].reshape(-1, 3), ref, rtol=0, atol=2e-5)
    gt = img["rgb"].clone() + 0.01
    out = render_image(model, cam, pose, T(pl_np[0]), rgb_gt=gt[4:12], row0=4, row1=12)
    assert out["rgb"].shape == (8, 32, 3) and torch.equal(out["rgb"], img["rgb"][4:12])
    assert abs(render_image(model, cam, pose, T(pl_np[0]), rgb_gt=gt)["psnr"] - 40.0) < 1e-2


def test_training_loop_descends():
    """A few real optimisation steps (HIP no-grad stages + autograd core + Adam): the loss on a fixed batch goes down
    and every parameter tensor moves."""
    from nrhints_amd.training import make_optimizer, train_step
    from nrhints_amd.synthetic import perturb_state
    torch.manual_seed(0)
    student = na.NeuSHintRenderer().cuda()
    st = perturb_state({k: v.detach().cpu().numpy().copy() for k, v in student.state_dict().items()})
    teacher = na.NeuSHintRenderer()
    teacher.load_state_dict({k: T(np.asarray(v)) for k, v in st.items()})
    teacher = teacher.cuda().eval()
    rays = make_rays(256, seed=5, spread=0.08)
    rb = _bundle(*rays)
    bg = torch.ones(1, 3).cuda()
    with torch.no_grad():
        gt = teacher(rb, background_rgb=bg).rgb
    before = {k: v.detach().clone() for k, v in student.named_parameters()}
    opt, sched = make_optimizer(student, lr=1e-3, warm_up_end=1)
    torch.manual_seed(1)
    losses = [train_step(student, rb, gt, bg, 50_000, opt, sched)["loss"] for _ in range(8)]
    assert all(np.isfinite(losses)) and losses[-1] < 0.8 * losses[1], losses
    moved = [k for k, v in student.named_parameters() if not torch.equal(v.detach(), before[k])]
    assert len(moved) == 46, len(moved)


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_off_default_branches(scene_states, prec):
    """pl-naive preset (no hints), Analytic normals, MaximalWeightPoint depth vs the imported reference
    (tests/golden/render_variants_b.npz), eval and one training-gradient smoke for the pl-naive model."""
    from nrhints_amd.synthetic import naive_state
    g = load_npz("render_variants_b.npz")
    rb = _bundle(*(g[k] for k in ("o", "d", "pl", "near", "far")))
    R = na.NeuSRendererConfig
    cfgs = {"pln": (R(shadow_hint=False, specular_hint=False), naive_state(scene_states["b"])),
            "ana": (R(normal_type=na.NormalComputationType.Analytic), scene_states["b"]),
            "mwp": (R(depth_type=na.DepthComputationType.MaximalWeightPoint), scene_states["b"])}
    for vt, (rcfg, st) in cfgs.items():
        model = na.NeuSHintRenderer(na.NeuSModelConfig(renderer=rcfg), precision=prec)
        model.load_state_dict({k: T(np.asarray(v)) for k, v in st.items()})
        model = model.cuda().eval()
        with torch.no_grad():
            out = model(rb, background_rgb=torch.ones(1, 3).cuda())
        np.testing.assert_allclose(out.rgb.cpu().numpy(), g[f"{vt}.rgb"], rtol=0, atol=1e-4)
        assert psnr(out.rgb.cpu().numpy(), g[f"{vt}.rgb"]) > 80.0
        np.testing.assert_allclose(out.depth.cpu().numpy(), g[f"{vt}.depth"], rtol=0, atol=3e-4 if vt != "mwp" else 2e-2)
        if vt == "mwp":   # argmax can legitimately flip between two near-equal weights: judged on the bulk
            assert np.mean(np.abs(out.depth.cpu().numpy() - g["mwp.depth"]) > 3e-4) < 0.05
        d = np.abs(out.weights.cpu().numpy() - g[f"{vt}.weights"])
        assert d.mean() < 3e-5
        if vt == "pln":
            assert out.visibilities is None and out.specular_cue is None
        else:
            np.testing.assert_allclose(out.visibilities.cpu().numpy(), g[f"{vt}.visibilities"], rtol=0, atol=3e-3)
    # gradients flow through the pl-naive model too
    model = na.NeuSHintRenderer(na.NeuSModelConfig(renderer=cfgs["pln"][0]), precision=prec)
    model.load_state_dict({k: T(np.asarray(v)) for k, v in cfgs["pln"][1].items()})
    model = model.cuda()
    out = model(rb, is_training=True, background_rgb=torch.ones(1, 3).cuda(), global_step=1000)
    out.rgb.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


@pytest.mark.parametrize("npts", [256, 1000])
def test_sdf_function_hip_backward(scene, npts):
    """Training forward (sdf_kernel<3>) + tangent / adjoint sweeps (csrc/nrh_sdf_train.hip) + rocBLAS dW GEMMs against
    the fp64 second-order autograd of the oracle (the reference's create_graph formulation, fields/sdf_field.py:136-148):
    gradients w.r.t. the points and all 40 raw SDF-network parameters through value, feature, d sdf/dx and an
    eikonal term.  The yardstick is the error of the same maths in fp32 torch ops on the GPU ("manual")."""
    from nrhints_amd.sdf_function import sdf_value_feat_grad
    from tests.torch_backends import sdf_value_feat_grad_manual
    tag, model, packed, _, p64 = scene
    rs = np.random.RandomState(5)
    pts_np = rs.uniform(-0.7, 0.7, size=(npts, 3))
    cs, cf, cg = rs.randn(npts, 1), rs.randn(npts, 256) * 0.1, rs.randn(npts, 3)

    def loss_of(sdf, feat, g, conv):
        return (sdf * conv(cs)).sum() + (feat * conv(cf)).sum() + (g * conv(cg)).sum() + ((g.norm(dim=-1) - 1) ** 2).sum()

    # fp64 reference on the CPU
    st64 = {k: v.detach().double().cpu() for k, v in model.state_dict().items()}
    leaves64 = {k: v.clone().requires_grad_(True) for k, v in st64.items() if k.startswith("sdf_network")}
    P64 = orc.params_from_state({**st64, **leaves64}, dtype=torch.float64)
    x64 = torch.tensor(pts_np, dtype=torch.float64, requires_grad=True)
    sdf64, feat64 = orc.sdf_forward(P64, x64)
    g64 = orc.sdf_gradient_autograd(P64, x64, create_graph=True)
    ref = torch.autograd.grad(loss_of(sdf64, feat64, g64, lambda a: torch.tensor(a)), [x64] + list(leaves64.values()))
    names = ["pts"] + list(leaves64)

    def run(impl):
        leaves = {k: p for k, p in model.named_parameters() if k.startswith("sdf_network")}
        dense = pk.dense_params(dict(model.named_parameters()))
        x = cu(pts_np.astype(np.float32)).requires_grad_(True)
        sdf, feat, g = sdf_value_feat_grad(dense, x, packed=packed) if impl == "hip" else sdf_value_feat_grad_manual(dense, x)
        grads = torch.autograd.grad(loss_of(sdf, feat, g, lambda a: cu(a.astype(np.float32))), [x] + [leaves[k] for k in names[1:]])
        return (sdf, feat, g), grads

    (sdf_h, feat_h, g_h), got = run("hip")
    _, base = run("manual")
    for a, b, tol in ((sdf_h, sdf64, 2e-6), (feat_h, feat64, 2e-5), (g_h, g64, 2e-4)):
        assert (a.detach().cpu().double() - b.detach()).abs().max().item() < tol * max(1.0, b.abs().max().item())
    for name, a, b, r in zip(names, got, base, ref):
        scale = r.abs().max().item() + 1e-30
        e_hip = (a.detach().cpu().double() - r).abs().max().item() / scale
        e_t32 = (b.detach().cpu().double() - r).abs().max().item() / scale
        assert e_hip <= 5.0 * e_t32 + 2e-5, (name, e_hip, e_t32)


def test_register_view_recovers_pose_delta(scene_states):
    """register_view (pipelines/base_pipeline.py:71-91) through the differentiable ray generator and the renderer's ray
    gradients: a view rendered from a shifted camera is registered by optimising ``cam_pose_adjustment``; the L1 loss
    must drop and the recovered translation must move towards the true shift."""
    from nrhints_amd import RawPixelBundle, RayGenerator, RayGeneratorConfig
    from nrhints_amd.pipeline import CameraModel
    from nrhints_amd.training import register_view
    st = scene_states["b"]
    model = na.NeuSHintRenderer(na.NeuSModelConfig())
    model.load_state_dict({k: T(np.asarray(v)) for k, v in st.items()})
    model = model.cuda().eval()
    H = W = 48
    cam = CameraModel(H=H, W=W, cx=W / 2, cy=H / 2, fx=60.0, fy=60.0)
    pos = torch.tensor([0.0, -3.5, 1.2])
    fwd = -pos / pos.norm()
    right = torch.linalg.cross(fwd, torch.tensor([0.0, 0.0, 1.0])); right = right / right.norm()
    up = torch.linalg.cross(right, fwd)
    pose = torch.eye(4); pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = right, up, -fwd, pos
    shift = torch.tensor([0.08, 0.0, -0.06])
    hh, ww = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")

    def bundle(p, rgb=None):
        return RawPixelBundle(img_indices=torch.zeros(H, W, 1, dtype=torch.long), h_indices=hh[..., None], w_indices=ww[..., None],
                              poses=p.expand(H, W, 4, 4), pls=torch.tensor([1.0, -3.0, 3.0]).expand(H, W, 3), rgb_gt=rgb)

    true_pose = pose.clone(); true_pose[:3, 3] += shift
    rg_true = RayGenerator(cam, 1, RayGeneratorConfig()).cuda()
    with torch.no_grad():
        gt = model(rg_true(bundle(true_pose).flatten().to("cuda")), background_rgb=torch.ones(1, 3).cuda()).rgb.reshape(H, W, 3).cpu()
    rg = RayGenerator(cam, 1, RayGeneratorConfig(cam_opt_mode="SO3xR3")).cuda()
    gen = torch.Generator().manual_seed(3)
    losses = register_view(model, rg, bundle(pose, gt), "cuda", steps=120, batch_size=1024, lr=4e-3, generator=gen)
    first, last = np.mean(losses[:10]), np.mean(losses[-10:])
    assert last < 0.6 * first, (first, last)
    t = rg.cam_pose_adjustment.detach().cpu()[0, :3]
    assert (t - shift).norm() < 0.6 * shift.norm(), (t, shift)
    # the renderer itself was not touched
    for k, v in model.state_dict().items():
        assert torch.equal(v.cpu(), T(np.asarray(st[k]))), k


def test_alpha_train_kernels_vs_autograd(scene):
    """AlphaWeightsNormalsHip (csrc/nrh_rays_train.hip: alpha, exclusive transmittance product, weights, unit normals and
    their adjoint) against the same expressions differentiated by torch autograd in fp64
    (models/neus_hint_model.py:339-356, :521-525, :584)."""
    from nrhints_amd.autograd_core import AlphaWeightsNormalsHip
    tag, model, packed, _, _ = scene
    rs = np.random.RandomState(11)
    n, T_ = 37, 128
    sdf = rs.uniform(-0.02, 0.05, size=(n * T_, 1))
    grad = rs.randn(n * T_, 3) * 0.7
    grad[5] = 0.0                                      # F.normalize clamp branch
    dirs = rs.randn(n, 3); dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    dists = rs.uniform(0.002, 0.03, size=(n, T_))
    var = float(model.deviation_network.variance.item())
    cw, cn = rs.randn(n, T_), rs.randn(n * T_, 3)
    for ca in (0.0, 0.4, 1.0):
        # fp64 autograd reference on the CPU
        t64 = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
        s_, g_, d_, v_ = t64(sdf), t64(grad), t64(dirs), t64(np.array(var))
        inv_s = torch.exp(v_ * 10.0).clip(1e-6, 1e6)
        alpha = orc.alpha_from(s_, g_, d_[:, None, :].expand(n, T_, 3).reshape(-1, 3), torch.tensor(dists).reshape(-1, 1), inv_s, ca)
        w_ref = alpha.reshape(n, T_) * orc.excl_cumprod_one_minus(alpha.reshape(n, T_))
        n_ref = torch.nn.functional.normalize(g_, dim=-1)
        ref = torch.autograd.grad((w_ref * torch.tensor(cw)).sum() + (n_ref * torch.tensor(cn)).sum(), [s_, g_, d_, v_])
        # HIP
        s2, g2, d2 = (cu(a.astype(np.float32)).requires_grad_(True) for a in (sdf, grad, dirs))
        v2 = torch.tensor(var, device="cuda", requires_grad=True)
        inv_s_f = float(torch.exp(v2.detach() * 10.0).clip(1e-6, 1e6).item())
        w, nh = AlphaWeightsNormalsHip.apply(s2, g2, d2, cu(dists.astype(np.float32)), v2, inv_s_f, ca)
        got = torch.autograd.grad((w * cu(cw.astype(np.float32))).sum() + (nh * cu(cn.astype(np.float32))).sum(), [s2, g2, d2, v2])
        assert (w.cpu().double() - w_ref.detach()).abs().max() < 2e-5
        assert (nh.cpu().double() - n_ref.detach()).abs().max() < 2e-6
        for name, a, b in zip(("sdf", "grad", "dirs", "variance"), got, ref):
            scale = b.abs().max().item() + 1e-30
            assert (a.cpu().double() - b).abs().max().item() / scale < 2e-3, (name, ca)


@pytest.mark.parametrize("hints", [True, False])
def test_color_net_hip_vs_torch(scene_states, hints):
    """ColorNetHip (training forward + adjoint sweep kernels of csrc/nrh_color.hip + split-K dW GEMMs) against the same
    network in torch ops with autograd (fields/reflectance_network.py:68-96), both on the GPU, yardstick fp64 on the CPU."""
    from nrhints_amd import autograd_core as ac
    from nrhints_amd.synthetic import naive_state
    from tests.torch_backends import _color_net_torch
    st = scene_states["b"] if hints else naive_state(scene_states["b"])
    cfg = na.NeuSModelConfig() if hints else na.NeuSModelConfig(renderer=na.NeuSRendererConfig(shadow_hint=False, specular_hint=False))
    rs = np.random.RandomState(3)
    n, T_ = 24, 128
    feat = rs.randn(n * T_, 256) * 0.3
    pts = rs.uniform(-0.8, 0.8, size=(n * T_, 3))
    nrm = rs.randn(n * T_, 3); nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    enc_w = 99 if hints else 54
    ray_enc = rs.uniform(-1, 1, size=(n, enc_w))
    cc = rs.randn(n * T_, 3)
    res = {}
    for prec in ("f32", "f16x3", "torch32", "torch64"):
        on_gpu = prec != "torch64"
        dt = torch.float64 if prec == "torch64" else torch.float32
        model = na.NeuSHintRenderer(cfg, precision=prec if prec in ("f32", "f16x3") else "f32")
        model.load_state_dict({k: T(np.asarray(v)) for k, v in st.items()})
        model = model.to(dt)
        if on_gpu:
            model = model.cuda()
        dev = "cuda" if on_gpu else "cpu"
        leaves = {k: p for k, p in model.named_parameters() if k.startswith("color_network")}
        dense = pk.dense_params(dict(model.named_parameters()))
        tt = lambda a: torch.tensor(a, dtype=dt, device=dev, requires_grad=True)
        f_, p_, n_, e_ = tt(feat), tt(pts), tt(nrm), tt(ray_enc)
        if prec in ("f32", "f16x3"):
            packed = model.packed_params(torch.device("cuda", torch.cuda.current_device()), dense=dense)
            col = ac.ColorNetHip.apply(f_, p_, n_, e_, packed, *[dense[f"col_w{l}"] for l in range(5)], *[dense[f"col_b{l}"] for l in range(5)])
        else:
            sizes = [27, 27, 9, 36] if hints else [27, 27]
            col = _color_net_torch(dense, f_, p_, n_, list(torch.split(e_, sizes, dim=1)), n, T_, hints).reshape(-1, 3)
        grads = torch.autograd.grad((col * torch.tensor(cc, dtype=dt, device=dev)).sum(), [f_, p_, n_, e_] + list(leaves.values()))
        res[prec] = (col.detach().cpu().double(), [g.detach().cpu().double() for g in grads], ["feat", "pts", "normal", "ray_enc"] + list(leaves))
    ref_col, ref_g, names = res["torch64"]
    for prec in ("f32", "f16x3"):
        col, g, _ = res[prec]
        assert (col - ref_col).abs().max() < 5e-6, prec
        for name, a, b32, r in zip(names, g, res["torch32"][1], ref_g):
            # a ReLU whose pre-activation is within rounding of 0 may flip between two fp32 evaluations and moves the
            # adjoint entries downstream of it by a finite amount: compare in the relative L2 norm
            e_hip = ((a - r).norm() / (r.norm() + 1e-30)).item()
            e_t32 = ((b32 - r).norm() / (r.norm() + 1e-30)).item()
            assert e_hip <= 5.0 * e_t32 + 1e-3, (prec, name, e_hip, e_t32)


def test_weight_norm_fold_hip(scene_states):
    """One-launch fold of the 15 weight-normed linears and its adjoint (csrc/nrh_fold.hip) against the torch expression
    v * g / ||v||_row and autograd (fields/sdf_field.py:81-82, old-style weight_norm, dim=0)."""
    model = na.NeuSHintRenderer(na.NeuSModelConfig())
    model.load_state_dict({k: T(np.asarray(v)) for k, v in scene_states["b"].items()})
    model = model.cuda()
    named = dict(model.named_parameters())
    ref = pk.dense_params(named)
    got = pk.dense_params_hip(named)
    rs = np.random.RandomState(0)
    keys = [k for k in ref if "_w" in k]
    probes = {k: cu(rs.randn(*ref[k].shape).astype(np.float32)) for k in keys}
    for k in ref:
        assert torch.allclose(got[k], ref[k], rtol=2e-6, atol=1e-7), k
    prm = [p for n, p in named.items() if n.endswith("weight_g") or n.endswith("weight_v")]
    g_ref = torch.autograd.grad(sum((ref[k] * probes[k]).sum() for k in keys), prm)
    g_got = torch.autograd.grad(sum((got[k] * probes[k]).sum() for k in keys if k != "col_w2"), prm, allow_unused=True)
    g_ref2 = torch.autograd.grad(sum((pk.dense_params(named)[k] * probes[k]).sum() for k in keys if k != "col_w2"), prm, allow_unused=True)
    for (n, _), a, b in zip([(n, p) for n, p in named.items() if n.endswith("weight_g") or n.endswith("weight_v")], g_got, g_ref2):
        if b is None:       # the layer left out of the loss: the one-launch adjoint writes zeros
            assert a is not None and float(a.abs().max()) == 0.0, n
        else:
            assert torch.allclose(a, b, rtol=2e-4, atol=1e-6 * float(b.abs().max())), n


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_c_abi_standalone_program(scene_states, prec, tmp_path):
    """examples/c_abi_render.cpp - a host program with no Python and no torch that links libnrhints_hip.so - renders a
    scene file and must reproduce the Python host's result bit for bit (same kernels, same packed buffers)."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from dump_scene import dump
    from nrhints_amd import _lib
    exe = os.path.join(os.path.dirname(_lib.LIB_PATH), "c_abi_render")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    model = na.NeuSHintRenderer(na.NeuSModelConfig(), precision=prec)
    model.load_state_dict({k: T(np.asarray(v)) for k, v in scene_states["b"].items()})
    model = model.cuda().eval()
    rays = make_rays(300, seed=4, spread=0.12)
    scene, out = str(tmp_path / "scene.bin"), str(tmp_path / "out.bin")
    n = dump(scene, model, rays)       # packed on the GPU, exactly the buffers model.forward() uses
    res = subprocess.run([exe, scene, out], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    raw = np.fromfile(out, dtype=np.float32)
    rgb, depth, vis = raw[:3 * n].reshape(n, 3), raw[3 * n:4 * n], raw[4 * n:5 * n]
    with torch.no_grad():
        o = model(_bundle(*rays), background_rgb=torch.ones(1, 3).cuda())
    np.testing.assert_array_equal(rgb, o.rgb.cpu().numpy())
    np.testing.assert_array_equal(depth, o.depth.cpu().numpy().reshape(-1))
    np.testing.assert_array_equal(vis, o.visibilities.cpu().numpy().reshape(-1))


@pytest.mark.parametrize("variant", ["pln", "ana"])
def test_training_variants_gradients_vs_oracle(scene_states, variant):
    """One training forward + backward of the off-default models through the HIP training kernels - pl-naive (no hints:
    316-wide reflectance input, MKB = 4 kernels, configs/main_config.py:67-76) and Analytic normals into the reflectance
    net (models/neus_hint_model.py:621-625) - against the oracle's second-order-autograd formulation on the CPU with the
    same recorded jitter: loss and the gradient of every parameter tensor (relative L2)."""
    from nrhints_amd.synthetic import naive_state
    from nrhints_amd.training import train_loss_dict
    R = na.NeuSRendererConfig
    if variant == "pln":
        rcfg, st, okw = R(shadow_hint=False, specular_hint=False), naive_state(scene_states["b"]), dict(hints=False)
    else:
        rcfg, st, okw = R(normal_type=na.NormalComputationType.Analytic), scene_states["b"], dict(analytic_normal=True)
    n = 24
    o, d, pl, near, far = make_rays(n, seed=9, spread=0.1)
    rs = np.random.RandomState(2)
    t_p, t_s, gt = T(rs.rand(n, 1).astype(np.float32)), T(rs.rand(n, 64).astype(np.float32)), T(rs.rand(n, 3).astype(np.float32))
    model = na.NeuSHintRenderer(na.NeuSModelConfig(renderer=rcfg), precision="f16x3")
    model.load_state_dict({k: T(np.asarray(v)) for k, v in st.items()})
    model = model.cuda()
    out = model(_bundle(o, d, pl, near, far), is_training=True, background_rgb=torch.ones(1, 3).cuda(), global_step=30000,
                _t_rand_primary=t_p.cuda(), _t_rand_shadow=t_s.cuda())
    loss = train_loss_dict(out, gt.cuda())["loss"]
    loss.backward()
    leaves = {k: T(np.asarray(v)).clone().requires_grad_(True) for k, v in st.items()}
    ref = orc.render_forward(orc.params_from_state(leaves), T(o), T(d), T(pl), T(near), T(far), background_rgb=torch.ones(1, 3),
                             is_training=True, global_step=30000, t_rand_primary=t_p, t_rand_shadow=t_s, mode="as_written",
                             differentiable=True, **okw)
    loss_ref, _, _ = orc.train_loss(ref, gt)
    loss_ref.backward()
    assert abs(loss.item() - loss_ref.item()) < 2e-4 * max(1.0, abs(loss_ref.item()))
    for name, prm in model.named_parameters():
        want = leaves[name].grad
        got = prm.grad.detach().cpu()
        tol = 0.15 if name == "deviation_network.variance" else 2e-2     # a 2.7e-6 scalar with heavy cancellation, see above
        assert float((got - want).norm() / (want.norm() + 1e-30)) < tol, name


def test_extract_geometry_of_the_initial_sphere(scene_states):
    """extract_geometry (models/neus_hint_model.py:753-758): the reference initialisation is roughly a sphere of radius
    0.5 (geometric init, fields/sdf_field.py:58-99) - the mesh from the HIP SDF grid + iso-surface extraction is closed
    and its vertices have |sdf| ~ 0 under the oracle."""
    model = na.NeuSHintRenderer(na.NeuSModelConfig())
    model.load_state_dict({k: T(np.asarray(v)) for k, v in scene_states["a"].items()})
    model = model.cuda().eval()
    v, f = model.extract_geometry(torch.tensor([-1.0, -1.0, -1.0]), torch.tensor([1.0, 1.0, 1.0]), resolution=64, threshold=0.0)
    assert v.shape[0] > 1000 and f.shape[0] > 2000 and f.max() < v.shape[0]
    r = np.linalg.norm(v, axis=1)
    assert 0.3 < r.min() and r.max() < 0.9          # a bumpy sphere around radius 0.5: the encoding inputs are not zero-weighted exactly
    sdf, _ = orc.sdf_forward(orc.params_from_state(scene_states["a"]), T(v.astype(np.float32)), want_feat=False)
    assert float(sdf.abs().max()) < 5e-3 and float(sdf.abs().mean()) < 5e-4   # linear interpolation on a 2/63 grid (oracle grid: 2.0e-3 / 1.6e-4)
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    _, cnt = np.unique(np.sort(e, axis=1), axis=0, return_counts=True)
    assert (cnt == 2).all()


def test_graphed_train_step(scene_states):
    """training.GraphedTrainStep: forward + loss + backward + Adam captured into one hipGraph and replayed.  The replayed
    step must train (loss on a fixed batch falls, held-out error against the teacher falls), follow the host-side schedule
    through device scalars (learning rate, cos-anneal ratio) and leave the renderer usable eagerly afterwards."""
    from nrhints_amd.training import GraphedTrainStep, lr_factor
    torch.manual_seed(0)
    student = na.NeuSHintRenderer(na.NeuSModelConfig())
    student.load_state_dict({k: T(np.asarray(v)) for k, v in scene_states["a"].items()})
    teacher = na.NeuSHintRenderer(na.NeuSModelConfig())
    teacher.load_state_dict({k: T(np.asarray(v)) for k, v in scene_states["b"].items()})
    student, teacher = student.cuda(), teacher.cuda().eval()
    bg = torch.ones(1, 3).cuda()
    n = 256
    rb = _bundle(*make_rays(n, seed=21, spread=0.08))
    rb_eval = _bundle(*make_rays(n, seed=22, spread=0.08))
    with torch.no_grad():
        gt, gt_eval = teacher(rb, background_rgb=bg).rgb, teacher(rb_eval, background_rgb=bg).rgb
        err0 = float((student(rb_eval, background_rgb=bg).rgb - gt_eval).abs().mean())
    before = {k: v.detach().clone() for k, v in student.state_dict().items()}
    step = GraphedTrainStep(student, n, bg, lr=5e-4, warm_up_end=20, global_step=30000)
    losses = [step(rb, gt, global_step=30000 + i)["loss"] for i in range(60)]
    assert all(np.isfinite(losses)) and np.mean(losses[-10:]) < 0.9 * np.mean(losses[:10]), (losses[:3], losses[-3:])
    assert abs(float(step.lr_t) - 5e-4 * lr_factor(30059, 20, 1_000_000, 0.05)) < 1e-9
    assert abs(float(student.dyn_scalars[1]) - 30059 / 50000) < 1e-6
    inv_s = float(torch.exp(student.deviation_network.variance.detach() * 10.0))
    assert abs(float(student.dyn_scalars[0]) - inv_s) < 2e-2 * inv_s        # one step behind the parameter at most
    with pytest.raises(ValueError):
        step(_bundle(*make_rays(n // 2, seed=1)), gt[: n // 2], global_step=1)
    step.release()
    assert student.dyn_scalars is None
    changed = sum(int(not torch.equal(before[k], v)) for k, v in student.state_dict().items())
    assert changed == len(before)
    with torch.no_grad():
        err1 = float((student(rb_eval, background_rgb=bg).rgb - gt_eval).abs().mean())
    assert err1 < err0, (err0, err1)
