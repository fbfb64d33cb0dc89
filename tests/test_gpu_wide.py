"""GPU parity of the wide f16x3 SDF kernels (csrc/nrh_sdf32.hip: 32-point tiles, one wave per SIMD, AGPR-resident activations)
through the C ABI (nrh_sdf_eval_wide): against the fp64 oracle, the golden fixtures the imported reference produced, and the
16-point kernels on the same inputs.  Tolerances are float32 ones, written next to each assertion."""
import numpy as np
import pytest
import torch

import nrhints_amd as na
from nrhints_amd import ops, packing as pk
from nrhints_amd.synthetic import make_rays
from oracle import neus_oracle as orc
from tests.conftest import load_npz

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def cu(a):
    return (T(a) if isinstance(a, np.ndarray) else a).float().contiguous().cuda()


@pytest.fixture(scope="module", params=["a", "b"])
def wscene(request, scene_states):
    st = scene_states[request.param]
    model = na.NeuSHintRenderer(na.NeuSModelConfig(), precision="f16x3")
    model.load_state_dict({k: T(np.asarray(v)) for k, v in st.items()})
    model = model.cuda().eval()
    packed = model.packed_params(torch.device("cuda", torch.cuda.current_device()))
    assert packed["sdf_w32"].dtype == torch.float16 and packed["sdf_tab32"].shape == (11, 256)
    return request.param, model, packed, orc.params_from_state(st, torch.float64)


def _at_points(mode, packed, pts):
    zeros = torch.zeros_like(pts)
    t = torch.zeros(pts.shape[0], dtype=torch.float32, device=pts.device)
    return ops.sdf_eval_wide(mode, packed["sdf_w32"], packed["sdf_tab32"], pts.contiguous(), zeros, t, 1)


@pytest.mark.parametrize("npts", [1, 31, 33, 128, 129, 1000, 4096 + 5, 40000])
def test_wide_sdf_modes_vs_oracle(wscene, npts):
    tag, model, packed, p64 = wscene
    g = torch.Generator().manual_seed(npts)
    pts = (torch.rand(npts, 3, generator=g) * 2 - 1) * 0.95
    o_sdf, o_feat, o_grad = orc.sdf_forward_grad_analytic(p64, pts.double())
    for mode in (0, 1, 2):
        sdf, grad, feat = _at_points(mode, packed, pts.cuda())
        # fp16 hi/lo products, fp32 accumulation, 8 layers of K = 256: same class as the fp32 chain (measured 8e-7)
        np.testing.assert_allclose(sdf.cpu().numpy()[:, 0], o_sdf.numpy()[:, 0], rtol=0, atol=5e-6)
        if mode >= 1:
            # unorm16 sigma' hand-off (7.6e-6 per layer) dominates; gradient magnitude ~1 (scene b up to ~3)
            np.testing.assert_allclose(grad.cpu().numpy(), o_grad.numpy(), rtol=0, atol=1e-4 if tag == "a" else 5e-4)
        if mode == 2:
            f = pk.feat_tiles_to_rows(feat.cpu(), npts).numpy()
            np.testing.assert_allclose(f, o_feat.numpy(), rtol=0, atol=3e-5)


def test_wide_sdf_golden_fixture(wscene):
    """Directly against what the imported reference produced (tests/golden/unit_*.npz)."""
    tag, model, packed, _ = wscene
    u = load_npz(f"unit_{tag}.npz")
    sdf, grad, feat = _at_points(2, packed, cu(u["sdf_pts"]))
    P = u["sdf_pts"].shape[0]
    out = np.concatenate([sdf.cpu().numpy(), pk.feat_tiles_to_rows(feat.cpu(), P).numpy()], axis=1)
    np.testing.assert_allclose(out, u["sdf_out_f64"], rtol=0, atol=3e-5)
    np.testing.assert_allclose(out, u["sdf_out"], rtol=0, atol=3e-5)
    np.testing.assert_allclose(grad.cpu().numpy(), u["sdf_grad_f64"], rtol=0, atol=1e-4 if tag == "a" else 5e-4)


def test_wide_sdf_along_rays_strided(wscene):
    """Ray-parametrised points with a row stride (the sampler's calling convention, 16 and 24 samples per ray)."""
    tag, model, packed, p64 = wscene
    o, d, pl, near, far = make_rays(50, seed=5, spread=0.1)
    for nper in (16, 24):
        z = np.zeros((50, 128), np.float32)
        z[:, :nper] = near + (far - near) * np.linspace(0, 1, nper, dtype=np.float32)[None]
        sdf, _, _ = ops.sdf_eval_wide(0, packed["sdf_w32"], packed["sdf_tab32"], cu(o), cu(d), cu(z), nper, t_stride=128)
        pts = (T(o)[:, None] + T(d)[:, None] * T(z[:, :nper])[..., None]).reshape(-1, 3)
        ref = orc.sdf_forward(p64, pts.double(), False)[0].reshape(50, nper)
        np.testing.assert_allclose(sdf.cpu().numpy(), ref.numpy(), rtol=0, atol=5e-6)


def test_wide_matches_16_point_kernels(wscene):
    """Same inputs through both f16x3 kernel families: two independent code paths, agreement at the float32 level."""
    tag, model, packed, _ = wscene
    g = torch.Generator().manual_seed(7)
    pts = ((torch.rand(3000, 3, generator=g) * 2 - 1) * 0.9).cuda()
    s32, g32, f32 = _at_points(2, packed, pts)
    s16, g16, f16 = ops.sdf_at_points(2, packed["sdf_w"], packed["sdf_b"], packed["sdf_head"], pts)
    np.testing.assert_allclose(s32.cpu().numpy(), s16.cpu().numpy(), rtol=0, atol=4e-6)
    np.testing.assert_allclose(g32.cpu().numpy(), g16.cpu().numpy(), rtol=0, atol=2e-4 if tag == "a" else 8e-4)
    np.testing.assert_allclose(f32.cpu().numpy(), f16.cpu().numpy(), rtol=0, atol=3e-5)


def test_wide_deterministic_and_repeatable(wscene):
    tag, model, packed, _ = wscene
    g = torch.Generator().manual_seed(11)
    pts = ((torch.rand(5000, 3, generator=g) * 2 - 1) * 0.9).cuda()
    a = _at_points(2, packed, pts)
    b = _at_points(2, packed, pts)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_render_wide_vs_16_point(wscene):
    """The whole evaluation render with the SDF network on the wide kernels vs on the 16-point kernels."""
    tag, model, packed, _ = wscene
    o, d, pl, near, far = make_rays(300, seed=3, spread=0.1)
    rb = na.RayBundle(origins=cu(o), directions=cu(d), pl_positions=cu(pl), nears=cu(near), fars=cu(far))
    bg = torch.ones(1, 3, device="cuda")
    with torch.no_grad():
        model.wide_kernels = True
        a = model(rb, is_training=False, background_rgb=bg)
        model.wide_kernels = False
        b = model(rb, is_training=False, background_rgb=bg)
        model.wide_kernels = True
    # both are fp32-class evaluations of the same network; the inverse-CDF sampler amplifies last-bit differences of the
    # sdf at isolated samples (see test_gpu_parity.py), rgb stays at the reference's own fp32-vs-fp64 noise
    assert float((a.rgb - b.rgb).abs().max()) < 5e-5
    assert float((a.depth - b.depth).abs().max()) < 3e-4
    assert float((a.visibilities - b.visibilities).abs().max()) < 3e-3


def test_fused_feature_head(wscene, scene_states):
    """NrhNet.feat_fused: the feature head multiplied into the feature block of the reflectance net's first layer at pack
    time (packing32.fuse_feature_head; both maps are linear: fields/sdf_field.py:119-123 -> reflectance_network.py:77-84).
    The mode-2 kernel with the fused streams returns W0feat * feature, and the evaluation render with the fused path equals
    the render without it to fp32 round-off (same sampler decisions: the SDF values do not change at all)."""
    from nrhints_amd import packing as pkg, packing32
    tag, model, packed, _ = wscene
    dev = torch.device("cuda", torch.cuda.current_device())
    d = pkg.dense_params_device({k: v.detach().float().to(dev) for k, v in model.state_dict().items()})     # the renderer's own fold
    w32f, tab32f = packing32.pack_sdf32_fused(d)
    o, dd, pl, near, far = make_rays(64, seed=12, spread=0.1)
    t = torch.rand(64, 128, device="cuda") * 2 + 2
    plain = ops.sdf_eval_wide(2, packed["sdf_w32"], packed["sdf_tab32"], cu(o), cu(dd), t, 128)
    fused = ops.sdf_eval_wide(2, w32f, tab32f, cu(o), cu(dd), t, 128)
    assert torch.equal(plain[0], fused[0]) and torch.equal(plain[1], fused[1])          # sdf, gradient: same blocks
    feat = pkg.feat_tiles_to_rows(plain[2], 64 * 128).double()
    want = feat @ d["col_w0"].double()[:, 60:316].t()
    got = pkg.feat_tiles_to_rows(fused[2], 64 * 128).double()
    assert float((got - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))
    rb = na.RayBundle(origins=cu(o), directions=cu(dd), pl_positions=cu(pl), nears=cu(near), fars=cu(far))
    bg = torch.ones(1, 3, device="cuda")
    with torch.no_grad():
        assert model.fuse_feature_head
        a = model(rb, is_training=False, background_rgb=bg)
        assert model._packed.get("sdf_w32f") is not None
        model.fuse_feature_head = False
        b = model(rb, is_training=False, background_rgb=bg)
        model.fuse_feature_head = True
    assert torch.equal(a.weights, b.weights) and torch.equal(a.depth, b.depth) and torch.equal(a.visibilities, b.visibilities)
    assert float((a.rgb - b.rgb).abs().max()) < 2e-6


def test_wide_reflectance_kernel(wscene):
    """csrc/nrh_color32.hip (the reflectance net on the wide machinery, behind NrhNet.col_w32): the evaluation render with it
    against the render with the 16-point reflectance kernel (same fused feature head) and against the one without either -
    the SDF side is untouched (weights, depth, visibilities bit-equal), rgb agrees to fp32 round-off; and the colours
    themselves against the fp64 oracle through the golden-fixture render test's tolerances elsewhere."""
    tag, model, packed, _ = wscene
    o, dd, pl, near, far = make_rays(257, seed=21, spread=0.1)
    rb = na.RayBundle(origins=cu(o), directions=cu(dd), pl_positions=cu(pl), nears=cu(near), fars=cu(far))
    bg = torch.ones(1, 3, device="cuda")
    with torch.no_grad():
        assert model.wide_color and model.fuse_feature_head
        a = model(rb, is_training=False, background_rgb=bg)
        assert model._packed.get("col_w32") is not None
        a2 = model(rb, is_training=False, background_rgb=bg)
        model.wide_color = False
        b = model(rb, is_training=False, background_rgb=bg)
        model.fuse_feature_head = False
        c = model(rb, is_training=False, background_rgb=bg)
        model.wide_color, model.fuse_feature_head = True, True
    assert torch.equal(a.rgb, a2.rgb)                                   # deterministic
    for x in (b, c):
        assert torch.equal(a.weights, x.weights) and torch.equal(a.depth, x.depth) and torch.equal(a.visibilities, x.visibilities)
        assert float((a.rgb - x.rgb).abs().max()) < 3e-6, float((a.rgb - x.rgb).abs().max())
    assert torch.isfinite(a.rgb).all()


@pytest.mark.parametrize("scale", [1.0, 0.05, 8.0])
def test_wide_reflectance_kernel_vs_oracle(wscene, scene_states, scale):
    """nrh_color_eval_wide against the fp64 oracle (fields/reflectance_network.py:68-96) on free inputs of several magnitudes:
    points on real rays, random unit normals, random features (-> part = W0feat * feature), random hints."""
    from nrhints_amd import packing as pkg, packing32
    from oracle import neus_oracle as orc
    tag, model, packed, _ = wscene
    dev = torch.device("cuda", torch.cuda.current_device())
    d = pkg.dense_params({k: v.detach().float().to(dev) for k, v in model.state_dict().items()})
    c32, ctab = packing32.pack_color32(d)
    N = 5
    o, dd, pl, near, far = make_rays(N, seed=31, spread=0.1)
    g = torch.Generator().manual_seed(7)
    tmid = torch.rand(N, 128, generator=g) * 2 + 2
    nhat = torch.nn.functional.normalize(torch.randn(N * 128, 3, generator=g), dim=-1)
    feat = torch.randn(N * 128, 256, generator=g) * 0.3 * scale
    vis = torch.rand(N, 1, generator=g)
    cue = torch.rand(N, 4, generator=g) * 2 * scale
    T = torch.from_numpy
    raymisc = torch.zeros(N + 1, pkg.RAYMISC_STRIDE)
    raymisc[:N, 0:27] = orc.nerf_encode(T(dd), 4)
    raymisc[:N, 27:54] = orc.nerf_encode(T(pl), 4)
    raymisc[:N, 54:63] = orc.nerf_encode(vis, 4)
    raymisc[:N, 63:99] = orc.nerf_encode(cue, 4)
    part = (feat.double() @ d["col_w0"].double().cpu()[:, 60:316].t()).float()
    col = ops.color_eval_wide(c32, ctab, pkg.rows_to_feat_tiles(part).cuda(), cu(o), cu(dd), tmid.cuda(), nhat.cuda().contiguous(),
                              raymisc.cuda())
    p64 = orc.params_from_state(scene_states[tag], torch.float64)
    pts = (T(o)[:, None] + T(dd)[:, None] * tmid[..., None]).reshape(-1, 3)
    rep = lambda x: x[:, None, :].expand(N, 128, x.shape[-1]).reshape(N * 128, -1)
    ref = orc.color_forward(p64, pts.double(), nhat.double(), rep(T(dd)).double(), feat.double(), rep(T(pl)).double(), rep(vis).double(), rep(cue).double())
    err = (col.cpu().double() - ref).abs()
    assert float(err.max()) < 5e-6, (float(err.max()), float(err.mean()), int(err.reshape(N, 128, 3).amax(-1).argmax()))


@pytest.mark.parametrize("npts_per_ray,nrays", [(128, 7), (16, 5), (1, 37)])
def test_wide_jvp_mode_vs_oracle_and_reverse_mode(wscene, npts_per_ray, nrays):
    """nrh_sdf_eval_wide mode 3 (value + derivative ALONG the ray in forward mode: 16 points + 16 tangents per tile, no sigma'
    scratch) against the fp64 oracle and against mode 1 (reverse mode): same sdf bit for bit, <rd, grad> equal to fp32 round-off,
    grad parallel to rd.  Ray directions are not unit length here."""
    tag, model, packed, p64 = wscene
    o, d, pl, near, far = make_rays(nrays, seed=5, spread=0.12)
    d = d * np.linspace(0.5, 1.7, nrays, dtype=np.float32)[:, None]
    t = torch.rand(nrays, npts_per_ray, device="cuda") * 2 + 1.5
    s1, g1, _ = ops.sdf_eval_wide(1, packed["sdf_w32"], packed["sdf_tab32"], cu(o), cu(d), t, npts_per_ray)
    s3, g3, _ = ops.sdf_eval_wide(3, packed["sdf_w32"], packed["sdf_tab32"], cu(o), cu(d), t, npts_per_ray)
    assert torch.equal(s1, s3)
    dd = cu(d)[:, None, :].expand(nrays, npts_per_ray, 3).reshape(-1, 3)
    cos1, cos3 = (g1 * dd).sum(-1), (g3 * dd).sum(-1)
    scale = float(g1.abs().max())
    assert float((cos1 - cos3).abs().max()) < 3e-5 * max(1.0, scale), float((cos1 - cos3).abs().max())
    assert float((torch.linalg.cross(g3, dd)).abs().max()) < 1e-6 * max(1.0, scale)            # parallel to the ray
    pts = (torch.from_numpy(o)[:, None] + torch.from_numpy(d)[:, None] * t.cpu()[..., None]).reshape(-1, 3).double()
    o_sdf, _, o_grad = orc.sdf_forward_grad_analytic(p64, pts)
    np.testing.assert_allclose(s3.cpu().numpy().reshape(-1), o_sdf.numpy()[:, 0], rtol=0, atol=5e-6)
    np.testing.assert_allclose(cos3.cpu().numpy(), (o_grad.numpy() * dd.cpu().numpy()).sum(-1), rtol=0, atol=5e-5 * max(1.0, scale))


def test_render_with_and_without_shadow_jvp(wscene):
    """The evaluation render with the shadow march's last evaluation in forward mode (NrhNet.shadow_jvp) against the reverse-mode
    one: everything upstream of the shadow march is bit-equal, visibilities agree to fp32 round-off of <dir, grad>."""
    tag, model, packed, _ = wscene
    o, dd, pl, near, far = make_rays(300, seed=23, spread=0.1)
    rb = na.RayBundle(origins=cu(o), directions=cu(dd), pl_positions=cu(pl), nears=cu(near), fars=cu(far))
    bg = torch.ones(1, 3, device="cuda")
    with torch.no_grad():
        assert not model.shadow_jvp          # off by default: measured 1.6 % slower per frame than reverse mode
        model.shadow_jvp = True
        a = model(rb, is_training=False, background_rgb=bg)
        model.shadow_jvp = False
        b = model(rb, is_training=False, background_rgb=bg)
    assert torch.equal(a.weights, b.weights) and torch.equal(a.depth, b.depth)
    assert float((a.visibilities - b.visibilities).abs().max()) < 2e-4, float((a.visibilities - b.visibilities).abs().max())
    assert float((a.rgb - b.rgb).abs().max()) < 2e-5


@pytest.mark.parametrize("level", ["residuals_subnormal", "all_subnormal"])
def test_wide_sdf_subnormal_activations(scene_states, level):
    """one layer's activations scaled so that EVERY fp16 residual the forward chain hands on is a subnormal
    (level 1), or so that the hi halves are subnormals as well and the residuals fall below the smallest one (level 2), with
    the next layer's gain raised so that these values decide the output (tests/stress_states.py).  The wide kernels ship the
    UNSCALED residual (csrc/gen_mlp32.py split_ops) and rely on v_cvt_pkrtz, v_fma_mix, v_accvgpr_write and the MFMA honouring
    fp16 subnormals end to end: a flush anywhere on that path shows as 6e-5 / 3e-4 in the sdf and 2e-4 / 8e-4 in the feature
    (tests/test_packing32_emulated.py::test_sdf32_subnormal_stress_emulated), against the 5e-6 / 3e-5 asserted here."""
    from tests.stress_states import subnormal_stress_state
    st = subnormal_stress_state(scene_states["b"], level)
    model = na.NeuSHintRenderer(na.NeuSModelConfig(), precision="f16x3")
    model.load_state_dict({k: T(np.asarray(v)) for k, v in st.items()})
    model = model.cuda().eval()
    packed = model.packed_params(torch.device("cuda", torch.cuda.current_device()))
    p64 = orc.params_from_state(st, torch.float64)
    g = torch.Generator().manual_seed(77)
    pts = (torch.rand(4096 + 19, 3, generator=g) * 2 - 1) * 0.9
    o_sdf, o_feat, o_grad = orc.sdf_forward_grad_analytic(p64, pts.double())
    for mode in (0, 1, 2):
        sdf, grad, feat = _at_points(mode, packed, pts.cuda())
        np.testing.assert_allclose(sdf.cpu().numpy()[:, 0], o_sdf.numpy()[:, 0], rtol=0, atol=5e-6)
        if mode >= 1:
            np.testing.assert_allclose(grad.cpu().numpy(), o_grad.numpy(), rtol=0, atol=5e-4)
        if mode == 2:
            np.testing.assert_allclose(pk.feat_tiles_to_rows(feat.cpu(), pts.shape[0]).numpy(), o_feat.numpy(), rtol=0, atol=3e-5)
    # the 16-point f16x3 kernels (scaled residual) on the same state: the second, independent code path
    s16, g16, f16 = ops.sdf_at_points(2, packed["sdf_w"], packed["sdf_b"], packed["sdf_head"], pts.cuda())
    np.testing.assert_allclose(s16.cpu().numpy()[:, 0], o_sdf.numpy()[:, 0], rtol=0, atol=5e-6)
    np.testing.assert_allclose(pk.feat_tiles_to_rows(f16.cpu(), pts.shape[0]).numpy(), o_feat.numpy(), rtol=0, atol=3e-5)


# ---- precision "f16": the one-term (single-pass) builds of the wide SDF kernels (csrc/nrh_wide1.hip) -------------------------------
@pytest.mark.parametrize("npts", [33, 1000, 40000])
def test_one_term_sdf_modes_vs_oracle(wscene, npts):
    """nrh_sdf_eval_wide_f16: the same packed streams through ONE fp16 MFMA per K step (weights and activations of the SDF network
    at 11 bits).  A reduced-precision mode: against the float64 oracle the sdf is good to a few 1e-4 (emulated before it was built:
    2-6e-4, profiles/r05/one_term_emulation.log) where the three-term kernels hold 5e-6 - and it must NOT be the three-term result
    (the test tells the two builds apart); deterministic; modes 0 / 1 / 2 / 3 agree on the sdf."""
    tag, model, packed, p64 = wscene
    g = torch.Generator().manual_seed(1000 + npts)
    pts = (torch.rand(npts, 3, generator=g) * 2 - 1) * 0.95
    o_sdf, o_feat, o_grad = orc.sdf_forward_grad_analytic(p64, pts.double())
    zeros = torch.zeros_like(pts).cuda()
    t = torch.zeros(npts, dtype=torch.float32, device="cuda")
    run = lambda mode, one: ops.sdf_eval_wide(mode, packed["sdf_w32"], packed["sdf_tab32"], pts.cuda().contiguous(), zeros, t, 1, one_term=one)
    ref3 = run(0, False)[0]
    outs = {}
    for mode in (0, 1, 2):
        sdf, grad, feat = run(mode, True)
        outs[mode] = sdf
        err = float(np.abs(sdf.cpu().numpy()[:, 0] - o_sdf.numpy()[:, 0]).max())
        assert err < 2e-3, (mode, err)
        if mode >= 1:
            assert float(np.abs(grad.cpu().numpy() - o_grad.numpy()).max()) < (3e-2 if tag == "a" else 1e-1)
        if mode == 2:
            assert float(np.abs(pk.feat_tiles_to_rows(feat.cpu(), npts).numpy() - o_feat.numpy()).max()) < 5e-3
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.equal(run(0, True)[0], outs[0])                                    # deterministic
    d3 = float((outs[0] - ref3).abs().max())
    assert d3 > 1e-6, d3                                                          # really the one-term arithmetic


@pytest.mark.parametrize("tag", ["a", "b"])
def test_one_term_render_vs_reference(scene_states, tag):
    """precision "f16" end to end (evaluation render: both samplers, render_core and the shadow march on the one-term kernels, the
    reflectance net and everything else as f16x3) against the reference's recorded render: PSNR(ours, reference) far above the
    50 dB SURVEY 8c / 8d require of a reduced-precision mode (|delta PSNR vs ground truth| < 0.05 dB at 30 dB), rgb within 5e-3; the
    default precision on the same rays is 30+ dB closer (so the mode is what it says); training with this precision runs the f16x3
    kernels (same gradients)."""
    from nrhints_amd.synthetic import psnr
    g = load_npz(f"render_{tag}.npz")
    mk = lambda prec: na.NeuSHintRenderer(na.NeuSModelConfig(), precision=prec)
    rb = na.RayBundle(origins=T(g["o"]).cuda(), directions=T(g["d"]).cuda(), pl_positions=T(g["pl"]).cuda(), nears=T(g["near"]).cuda(),
                      fars=T(g["far"]).cuda())
    outs = {}
    for prec in ("f16", "f16x3"):
        m = mk(prec)
        m.load_state_dict({k: T(np.asarray(v)) for k, v in scene_states[tag].items()})
        m = m.cuda().eval()
        with torch.no_grad():
            outs[prec] = m(rb, background_rgb=torch.ones(1, 3).cuda())
    ref = g["rgb_f64"]
    p1, p3 = psnr(outs["f16"].rgb.cpu().numpy(), ref), psnr(outs["f16x3"].rgb.cpu().numpy(), ref)
    assert p1 > 60.0 and p3 > p1 + 20.0, (p1, p3)
    assert float(np.abs(outs["f16"].rgb.cpu().numpy() - ref).max()) < 5e-3
    assert float(np.abs(outs["f16"].depth.cpu().numpy() - g["depth_f64"]).max()) < 3e-2
    assert float(np.abs(outs["f16"].visibilities.cpu().numpy() - g["visibilities_f64"]).max()) < 1e-2
    assert outs["f16"].weights.shape == (96, 128) and bool(torch.isfinite(outs["f16"].rgb).all())
