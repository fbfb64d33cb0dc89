"""GPU parity of the channel-split SDF kernel for small point sets (csrc/nrh_sdf_split.hip, C entry nrh_sdf_eval_split): bit-identical
to the 16-point f16x3 kernel nrh_sdf_eval(precision 1, mode 0) on the same packed parameters, within float32 tolerance of the
float64 oracle and of the golden fixtures the imported reference produced, and taken by the training forward's sampler."""
import numpy as np
import pytest
import torch

import nrhints_amd as na
from nrhints_amd import ops
from nrhints_amd.synthetic import make_rays
from oracle import neus_oracle as orc
from tests.conftest import load_npz

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def cu(a):
    return (T(a) if isinstance(a, np.ndarray) else a).float().contiguous().cuda()


@pytest.fixture(scope="module", params=["a", "b"])
def sscene(request, scene_states):
    st = scene_states[request.param]
    model = na.NeuSHintRenderer(na.NeuSModelConfig(), precision="f16x3")
    model.load_state_dict({k: T(np.asarray(v)) for k, v in st.items()})
    model = model.cuda().eval()
    packed = model.packed_params(torch.device("cuda", torch.cuda.current_device()))
    return request.param, model, packed, orc.params_from_state(st, torch.float64)


# (rays, samples per ray, row stride): the sampler's two calling conventions, ragged tails (points not a multiple of 16 / 32),
# one point, and a set larger than one round of workgroups
SHAPES = [(64, 64, 128), (64, 16, 16), (37, 16, 16), (1, 1, 1), (3, 5, 8), (128, 64, 128), (1000, 16, 16), (300, 64, 128)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("tiles", [0, 1, 2])
def test_split_bit_identical_to_the_16_point_kernel(sscene, shape, tiles):
    tag, model, packed, p64 = sscene
    n, nper, stride = shape
    o, d, pl, near, far = make_rays(n, seed=11 + n, spread=0.1)
    z = np.zeros((n, stride), np.float32)
    z[:, :nper] = near + (far - near) * np.linspace(0, 1, nper, dtype=np.float32)[None]
    args = (packed["sdf_w"], packed["sdf_b"], packed["sdf_head"], cu(o), cu(d), cu(z), nper)
    ref, _, _ = ops.sdf_eval(0, *args, t_stride=stride)
    got = ops.sdf_eval_split(*args, t_stride=stride, tiles=tiles)
    assert got.shape == ref.shape == (n, nper)
    assert torch.equal(got, ref), f"max |diff| {float((got - ref).abs().max()):.3e}"          # bit for bit
    again = ops.sdf_eval_split(*args, t_stride=stride, tiles=tiles)
    assert torch.equal(got, again)
    pts = (T(o)[:, None] + T(d)[:, None] * T(z[:, :nper])[..., None]).reshape(-1, 3)
    o_sdf = orc.sdf_forward(p64, pts.double(), False)[0].reshape(n, nper)
    # fp16 hi/lo products, fp32 accumulation, 8 layers of K = 256: the float32 chain's class (the wide kernels' bound)
    np.testing.assert_allclose(got.cpu().numpy(), o_sdf.numpy(), rtol=0, atol=5e-6)


def test_split_golden_fixture(sscene):
    """Directly against what the imported reference produced (tests/golden/unit_*.npz)."""
    tag, model, packed, _ = sscene
    u = load_npz(f"unit_{tag}.npz")
    pts = cu(u["sdf_pts"])
    t = torch.zeros(pts.shape[0], dtype=torch.float32, device=pts.device)
    sdf = ops.sdf_eval_split(packed["sdf_w"], packed["sdf_b"], packed["sdf_head"], pts, torch.zeros_like(pts), t, 1)
    np.testing.assert_allclose(sdf.cpu().numpy()[:, 0], u["sdf_out_f64"][:, 0], rtol=0, atol=5e-6)
    np.testing.assert_allclose(sdf.cpu().numpy()[:, 0], u["sdf_out"][:, 0], rtol=0, atol=5e-6)


def test_split_rejects_bad_arguments(sscene):
    tag, model, packed, _ = sscene
    o, d, pl, near, far = make_rays(4, seed=1, spread=0.1)
    z = cu(np.zeros((4, 16), np.float32))
    with pytest.raises(RuntimeError, match="tiles"):
        ops.sdf_eval_split(packed["sdf_w"], packed["sdf_b"], packed["sdf_head"], cu(o), cu(d), z, 16, tiles=3)
    with pytest.raises(RuntimeError, match="stride"):
        ops.sdf_eval_split(packed["sdf_w"], packed["sdf_b"], packed["sdf_head"], cu(o), cu(d), z, 16, t_stride=8)


def test_training_sampler_takes_the_split_kernel(sscene):
    """nrh_render_forward_train with a 64-ray batch (the reference's per-rank share, trainer/trainer.py:116-123): its sampler
    passes (4 096 and 1 024 points) run on the split kernel, the evaluation path's on the wide kernels - same arithmetic class,
    so the two renderings of the same rays agree to the float32 noise of the sample placement."""
    tag, model, packed, p64 = sscene
    o, d, pl, near, far = make_rays(64, seed=77, spread=0.08)
    rb = na.RayBundle(origins=cu(o), directions=cu(d), pl_positions=cu(pl), nears=cu(near), fars=cu(far))
    bg = torch.ones(1, 3, device="cuda")
    out_t = model(rb, False, bg)                      # grad mode on, parameters require grad: the training forward
    assert out_t.rgb.requires_grad
    with torch.no_grad():
        out_e = model(rb, False, bg)
    np.testing.assert_allclose(out_t.rgb.detach().cpu().numpy(), out_e.rgb.cpu().numpy(), rtol=0, atol=5e-3)
    np.testing.assert_allclose(out_t.depth.detach().cpu().numpy(), out_e.depth.cpu().numpy(), rtol=0, atol=5e-3)
    w_t, w_e = out_t.weights.detach().cpu().numpy(), out_e.weights.cpu().numpy()
    assert np.mean(np.abs(w_t - w_e) < 1e-3) > 0.995


@pytest.mark.parametrize("prec,small", [("f16x3", 8192), ("f16x3", 16384), ("f32", 8192)])
def test_small_batch_builds_of_the_training_kernels_are_bit_identical(scene_states, prec, small):
    """The small-batch forms of the SDF training forward and its two backward sweeps: the channel-split kernels
    (csrc/nrh_sdf_train_split.hip: f16x3, at most 2 tiles per CU = 8 192 points on 256 CUs) and the 4-wave builds
    (csrc/nrh_small.hip: both precisions, at most 4 tiles per CU = 16 384 points).  A tile's arithmetic does not depend on how its
    work is spread: the same points evaluated alone and as the head of a 4x larger call (8-wave builds) give the same bits in every
    output and saved array."""
    st = scene_states["b"]
    model = na.NeuSHintRenderer(na.NeuSModelConfig(), precision=prec)
    model.load_state_dict({k: T(np.asarray(v)) for k, v in st.items()})
    pk = model.cuda().eval().packed_params(torch.device("cuda", torch.cuda.current_device()))
    g = torch.Generator().manual_seed(3)
    big = 4 * small
    pts = ((torch.rand(big, 3, generator=g) * 2 - 1) * 0.9).cuda()
    s_sdf, s_feat, s_grad, s_sv = ops.sdf_train_forward(pk["sdf_w"], pk["sdf_b"], pk["sdf_head"], pts[:small].contiguous())
    b_sdf, b_feat, b_grad, b_sv = ops.sdf_train_forward(pk["sdf_w"], pk["sdf_b"], pk["sdf_head"], pts)
    assert torch.equal(s_sdf, b_sdf[:small]) and torch.equal(s_feat, b_feat[:small]) and torch.equal(s_grad, b_grad[:small])
    for k in ("h", "s1", "t"):
        assert torch.equal(s_sv[k], b_sv[k][:, :small]), k
    assert torch.equal(s_sv["ge"][:, :112], b_sv["ge"][:small, :112])
    sbar, fbar, gbar = (torch.randn(big, generator=g).cuda(), torch.randn(big, 256, generator=g).cuda() * 0.1,
                        torch.randn(big, 3, generator=g).cuda())
    zeros3, t0 = torch.zeros(big, 3, device="cuda"), torch.zeros(big, 1, device="cuda")
    rs = ops.sdf_train_backward(pk["sdf_w"], pk["sdf_wt_feat"], pk["sdf_head"], pts[:small].contiguous(), zeros3[:small].contiguous(),
                                t0[:small].contiguous(), 1, s_sv, sbar[:small].contiguous(), fbar[:small].contiguous(), gbar[:small].contiguous())
    rb = ops.sdf_train_backward(pk["sdf_w"], pk["sdf_wt_feat"], pk["sdf_head"], pts, zeros3, t0, 1, b_sv, sbar, fbar, gbar)
    for k in ("abar", "coup", "zbar"):
        assert torch.equal(rs[k], rb[k][:, :small]), k
    assert torch.equal(rs["gebar"], rb["gebar"][:small]) and torch.equal(rs["pbar"], rb["pbar"][:small])


@pytest.mark.parametrize("shape", [(64, 128, 128), (37, 16, 16), (1, 1, 1), (3, 5, 8), (130, 128, 128)])
def test_split_gradient_kernel_bit_identical_to_the_16_point_kernel(sscene, shape):
    """nrh_sdf_grad_split (sdf + d sdf / dx, sigma' in the producing wave's registers) against nrh_sdf_eval(precision 1, mode 1) on the
    same packed parameters: the same bits, ragged tails included; and against the float64 oracle at the wide kernels' bounds."""
    tag, model, packed, p64 = sscene
    n, nper, stride = shape
    o, d, pl, near, far = make_rays(n, seed=23 + n, spread=0.1)
    z = np.zeros((n, stride), np.float32)
    z[:, :nper] = near + (far - near) * np.linspace(0, 1, nper, dtype=np.float32)[None]
    args = (packed["sdf_w"], packed["sdf_b"], packed["sdf_head"], cu(o), cu(d), cu(z), nper)
    r_sdf, r_grad, _ = ops.sdf_eval(1, *args, t_stride=stride)
    g_sdf, g_grad = ops.sdf_grad_split(*args, t_stride=stride)
    assert torch.equal(g_sdf, r_sdf), f"max |diff| {float((g_sdf - r_sdf).abs().max()):.3e}"
    assert torch.equal(g_grad, r_grad), f"max |diff| {float((g_grad - r_grad).abs().max()):.3e}"
    pts = (T(o)[:, None] + T(d)[:, None] * T(z[:, :nper])[..., None]).reshape(-1, 3)
    o_sdf, _, o_grad = orc.sdf_forward_grad_analytic(p64, pts.double())
    np.testing.assert_allclose(g_sdf.cpu().numpy().reshape(-1), o_sdf.numpy()[:, 0], rtol=0, atol=5e-6)
    # unorm16 sigma' hand-off (7.6e-6 per layer) dominates; gradient magnitude ~1 (scene b up to ~3)
    np.testing.assert_allclose(g_grad.cpu().numpy(), o_grad.numpy(), rtol=0, atol=1e-4 if tag == "a" else 5e-4)


@pytest.mark.parametrize("nrays", [64, 77, 128, 257, 1024])
def test_fused_sampler_step_is_bit_identical(sscene, nrays):
    """Small training batches run each per-ray sampler step in the TAIL of the SDF pass that feeds it (sdf_split_kernel: tile i of
    a 16-samples-per-ray pass is ray i's new samples; nrh_step.h) - one launch instead of two, 8 fewer per step.  Same bits as the
    two-launch form in every product of the training forward: sample positions, sections, weights, visibility, sdf, gradient - at
    one tile per workgroup (64, 77 rays), two (128, 257: an odd tile count leaves a workgroup's second tile empty) and at the
    largest batch that takes the split kernel (1 024 rays = 16 384 points per pass)."""
    from nrhints_amd import _lib
    tag, model, packed, p64 = sscene
    lib = _lib.load()
    rs = np.random.RandomState(nrays)
    o, d, pl, near, far = make_rays(nrays, seed=90 + nrays, spread=0.1)
    tp, ts = cu(rs.rand(nrays).astype(np.float32)), cu(rs.rand(nrays, 64).astype(np.float32))

    def forward():
        r = model._render_train(cu(o), cu(d), cu(pl), cu(near).reshape(-1), cu(far).reshape(-1), 0.6, tp, ts, 0)
        torch.cuda.synchronize()
        return {k: r[k].clone() for k in ("mid_z", "dists", "weights", "visibilities", "depth", "normals", "cue")} | {"sdf": r["pre"]["sdf"].clone()}

    was = lib.nrh_sampler_fusion(2)          # 2: fused wherever the kernel supports it (the default policy stops at one ray per CU)
    try:
        fused = forward()
        assert lib.nrh_sampler_fusion(1) == 2
        default = forward()
        assert lib.nrh_sampler_fusion(0) == 1
        plain = forward()
    finally:
        lib.nrh_sampler_fusion(was)
    for k in fused:
        assert torch.equal(fused[k], plain[k]), (k, float((fused[k] - plain[k]).abs().max()))
        assert torch.equal(default[k], plain[k]), k
    assert float(fused["weights"].sum()) > 0.1 * nrays          # (rays that hit something: the comparison is not vacuous)


@pytest.mark.parametrize("hints", [True, False])
def test_split_reflectance_kernels_bit_identical_to_the_8_wave_kernels(scene_states, hints):
    """csrc/nrh_color_split.hip: the reflectance net's training forward and adjoint sweep for small batches (one tile per workgroup,
    a stage's channels over its four waves) return the 8-wave kernels' bits - colour, save_h, save_misc; zbar, fbar, mbar - for
    the hinted model (105 -> 128 misc inputs) and the pl-naive one (60 -> 64).  64 rays = 8 192 points evaluated alone (split
    kernels) against the head of a 256-ray call on the same inputs (8-wave kernels; points are independent)."""
    from nrhints_amd import _lib, packing
    from nrhints_amd.synthetic import naive_state
    lib = _lib.load()
    P_ = _lib.ptr
    cfg = na.NeuSModelConfig() if hints else na.NeuSModelConfig(renderer=na.NeuSRendererConfig(shadow_hint=False, specular_hint=False))
    st = scene_states["b"] if hints else naive_state(scene_states["b"])
    model = na.NeuSHintRenderer(cfg, precision="f16x3")
    model.load_state_dict({k: T(np.asarray(v)) for k, v in st.items()})
    model = model.cuda()
    dev = torch.device("cuda", torch.cuda.current_device())
    d = packing.dense_params_device({k: v.detach().float() for k, v in model.state_dict().items()})
    pk = model.packed_params(dev, dense=d)
    cw, cb, cwt = pk["col_w"], pk["col_b"], pk["col_wt"]
    mw = 128 if hints else 64
    g = torch.Generator().manual_seed(3 + int(hints))
    n_big, n_small = 256, 64
    Pb, Ps = n_big * 128, n_small * 128
    feat = (torch.randn(Pb, 256, generator=g) * 0.3).cuda()
    pts = (torch.rand(Pb, 3, generator=g) * 2 - 1).cuda()
    nrm = torch.nn.functional.normalize(torch.randn(Pb, 3, generator=g), dim=-1).cuda()
    raymisc = (torch.rand(n_big, packing.RAYMISC_STRIDE, generator=g) * 2 - 1).cuda()
    zbar4 = (torch.randn(Pb, 3, generator=g) * 1e-3).cuda()
    new = lambda *s: torch.full(s, float("nan"), dtype=torch.float32, device="cuda")

    def run(n):
        P = n * 128
        color, save_h, save_misc = new(P, 3), new(4, P, 256), new(P, mw)
        _lib.check(lib.nrh_color_train_forward(1, int(hints), P_(cw, cw.dtype), P_(cb), P_(feat[:P].contiguous()), P_(pts[:P].contiguous()),
                                               P_(nrm[:P].contiguous()), P_(raymisc[:n].contiguous()), n, P_(color), P_(save_h), P_(save_misc),
                                               _lib.stream_handle()), "nrh_color_train_forward")
        zbar, fbar, mbar = new(4, P, 256), new(P, 256), new(P, mw)
        _lib.check(lib.nrh_color_train_backward(1, int(hints), P_(cwt, cwt.dtype), P_(zbar4[:P].contiguous()), P_(save_h), n, P_(zbar), P_(fbar),
                                                P_(mbar), 8.0, _lib.stream_handle()), "nrh_color_train_backward")
        torch.cuda.synchronize()
        return dict(color=color, save_h=save_h, save_misc=save_misc, zbar=zbar, fbar=fbar, mbar=mbar)

    big, small = run(n_big), run(n_small)
    for k, v in small.items():
        assert torch.isfinite(v).all(), k
        ref = big[k][:, :Ps] if v.dim() == 3 else big[k][:Ps]
        assert torch.equal(v, ref), (k, float((v - ref).abs().max()))
    assert float(small["color"].std()) > 1e-3 and float(small["zbar"].abs().max()) > 0       # not vacuous
