"""GPU tests of the autograd-free training step (nrhints_amd/train_fused.py) and of its new kernels, each against an
independent witness: the split-K bf16x3 weight-gradient GEMMs (csrc/nrh_dw.hip) against float64 products, the composite + loss +
seed kernels against torch autograd of the reference's loss expressions, the extended alpha adjoint against autograd, and the
whole step against the gradients the imported reference recorded (tests/golden/train_*.npz) and against the autograd path."""
import numpy as np
import pytest
import torch

import nrhints_amd as na
from nrhints_amd import _lib, dw, train_fused
from nrhints_amd.synthetic import make_rays
from oracle import neus_oracle as orc
from tests.conftest import grad_bound, load_npz

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def cu(a):
    return (T(a) if isinstance(a, np.ndarray) else a).float().contiguous().cuda()


def _bundle(o, d, pl, near, far):
    return na.RayBundle(origins=cu(o), directions=cu(d), pl_positions=cu(pl), nears=cu(near), fars=cu(far))


def _model(state, prec="f16x3", cfg=None):
    m = na.NeuSHintRenderer(cfg or na.NeuSModelConfig(), precision=prec)
    m.load_state_dict({k: T(np.asarray(v)) for k, v in state.items()})
    return m.cuda()


def _wide_range(rs, P, C, scale):
    """adjoint-like data: heavy-tailed per point and per channel (what an fp16 split could not hold without a per-tensor scale)"""
    return (rs.randn(P, C) * np.exp(rs.randn(P, 1) * 2.0) * np.exp(rs.randn(1, C) * 1.5) * scale).astype(np.float32)


@pytest.mark.parametrize("tiled", [False, True])
@pytest.mark.parametrize("P", [32, 2048, 131072 + 32 * 7])
def test_dw_gemm_vs_float64(P, tiled):
    """Every job shape of the training step in one nrh_dw_gemm call: two-pair full products, the 39-column layer-0 product, a
    one-column product with a column sum, a row-limited and scaled product, column maps, the transposed 3-column product; against
    float64 products of the same float32 data.  Error model of the bf16x3 split: operands rounded to 16 mantissa bits, products
    exact, fp32 accumulation: |err| <= 2^-16 sum_p |a b| worst case, a few 1e-6 of the entry in practice (asserted: 3e-5 of
    sum |a b|, and 3e-5 of the matrix scale).  ``tiled``: the 256-channel operands in the tiled layout of the training arrays
    (NrhDwJob.tiled_a / tiled_b: the weight-gradient kernel rotates each block's rows on their way into LDS), in every path of the
    kernel - full products, narrow ones, the thin matrix-vector path - and one job whose two pairs differ in layout; same results."""
    rs = np.random.RandomState(P % 1000)
    A1, A2 = cu(_wide_range(rs, P, 256, 1e-6)), cu(_wide_range(rs, P, 256, 1e-3))
    B1 = cu(np.log1p(np.exp(rs.randn(P, 256).astype(np.float32) * 3)) * 0.1)          # activation-like
    B2 = cu(_wide_range(rs, P, 256, 1e-4))
    E, GE = cu(rs.randn(P, 64).astype(np.float32)), cu(_wide_range(rs, P, 64, 1e-5))
    sb = cu(_wide_range(rs, P, 1, 1e-4)[:, 0])
    M3 = cu(_wide_range(rs, P, 3, 1e-2))
    misc = cu(rs.randn(P, 128).astype(np.float32))
    new = lambda *s: torch.full(s, float("nan"), dtype=torch.float32, device="cuda")
    out = dict(full=new(256, 256), bfull=new(256), l0=new(256, 39), bl0=new(256), rows=new(217, 256), brows=new(217), ws=new(1, 256), bs=new(1),
               w0=new(256, 361), b0=new(256), w4=new(3, 256), b4=new(3))
    fi, mi = dw.color_col_maps(torch.device("cuda"), True)
    out["mixed"] = new(256, 256)
    tl = (lambda x: dw.to_tiled(x)) if tiled else (lambda x: x)
    A1t, A2t, B1t, B2t = tl(A1), tl(A2), tl(B1), tl(B2)
    assert not tiled or (torch.equal(dw.from_tiled(A1t), A1) and not torch.equal(A1t, A1))
    t_ = bool(tiled)
    jobs = [dw.Job([A1t, A2t], [B1t, B2t], 256, 256, out["full"], colsum_a=out["bfull"], tiled_a=(t_, t_), tiled_b=(t_, t_)),
            dw.Job([A1t, A2t], [E, GE], 256, 39, out["l0"], colsum_a=out["bl0"], tiled_a=(t_, t_)),
            dw.Job([A2t], [B1t], 256, 256, out["rows"], rows=217, scale=2.0 ** -0.5, colsum_a=out["brows"], tiled_a=(t_,), tiled_b=(t_,)),
            dw.Job([B1t, A2t], [sb.reshape(P, 1), dw.ones(P, "cuda").reshape(P, 1)], 256, 1, out["ws"], transpose=True, scale=1.0 / 3.0,
                   colsum_b=out["bs"], scale_b=1.0 / 3.0, tiled_a=(t_, t_)),
            dw.Job([A1], [B1t], 256, 256, out["w0"], col_map=fi, colsum_a=out["b0"], tiled_b=(t_,)),
            dw.Job([A1], [misc], 256, 105, out["w0"], col_map=mi),
            dw.Job([B1t], [M3], 256, 3, out["w4"], transpose=True, colsum_b=out["b4"], tiled_a=(t_,)),
            dw.Job([A1t, A2], [B1, B2t], 256, 256, out["mixed"], tiled_a=(t_, False), tiled_b=(False, t_))]      # pairs of different layout
    dw.run(jobs, P)
    torch.cuda.synchronize()
    d = lambda t: t.double().cpu()
    a1, a2, b1, b2, e, ge, s_, m3, mc = (d(x) for x in (A1, A2, B1, B2, E, GE, sb, M3, misc))

    def check(got, ref, absref, what):
        err = (d(got) - ref).abs()
        assert bool(torch.isfinite(d(got)).all()), what
        assert float((err / (absref + 1e-300)).max()) < 3e-5, (what, float((err / (absref + 1e-300)).max()))
        assert float(err.max()) < 3e-5 * float(ref.abs().max()), (what, float(err.max()), float(ref.abs().max()))

    check(out["full"], a1.t() @ b1 + a2.t() @ b2, a1.abs().t() @ b1.abs() + a2.abs().t() @ b2.abs(), "two-pair product")
    check(out["mixed"], a1.t() @ b1 + a2.t() @ b2, a1.abs().t() @ b1.abs() + a2.abs().t() @ b2.abs(), "two-pair product, pairs of different layout")
    check(out["l0"], a1.t() @ e[:, :39] + a2.t() @ ge[:, :39], a1.abs().t() @ e[:, :39].abs() + a2.abs().t() @ ge[:, :39].abs(), "39 columns")
    check(out["rows"], (a2.t() @ b1)[:217] * 2.0 ** -0.5, (a2.abs().t() @ b1.abs())[:217], "217 rows, scaled")
    check(out["ws"], ((b1.t() @ s_[:, None] + a2.sum(0)[:, None]) / 3.0).t(), ((b1.abs().t() @ s_.abs()[:, None] + a2.abs().sum(0)[:, None]) / 3).t(), "head")
    w0 = torch.zeros(256, 361, dtype=torch.float64)
    w0[:, fi.cpu().long()] = a1.t() @ b1
    w0[:, mi.cpu().long()] = a1.t() @ mc[:, :105]
    w0a = torch.zeros(256, 361, dtype=torch.float64)
    w0a[:, fi.cpu().long()] = a1.abs().t() @ b1.abs()
    w0a[:, mi.cpu().long()] = a1.abs().t() @ mc[:, :105].abs()
    check(out["w0"], w0, w0a, "column maps")
    check(out["w4"], m3.t() @ b1, m3.abs().t() @ b1.abs(), "transposed 3 columns")
    # column sums are plain fp32 sums in a fixed order
    for got, ref, ab in ((out["bfull"], a1.sum(0), a1.abs().sum(0)), (out["brows"], a2.sum(0)[:217], a2.abs().sum(0)[:217]),
                         (out["bs"], s_.sum(0, keepdim=True) / 3, s_.abs().sum(0, keepdim=True) / 3), (out["b4"], m3.sum(0), m3.abs().sum(0))):
        assert float(((d(got) - ref).abs() / ab).max()) < 2e-6
    # deterministic: the same call again gives the same bits
    first = {k: v.clone() for k, v in out.items()}
    dw.run(jobs, P)
    for k in out:
        assert torch.equal(first[k], out[k]), k


@pytest.mark.parametrize("P", [32, 96, 2048 + 32, 131072 + 32 * 7])
def test_dw_gemm_half_operands(P):
    """NrhDwJob.half_ops: the four operands of a full product as float16 arrays in the half-tiled layout (what the f16x3 sweeps hand
    over in 16-bit mode), one fp16 MFMA pass, fragments by ds_read_b64_tr_b16 straight out of the LDS-DMA image.  Against float64
    products of the SAME float16 data (the operand rounding is the caller's, priced in profiles/dw16_emulation.py): products of
    fp16 values are exact in fp32, so what is left is fp32 accumulation - 2e-6 of sum |a b|.  With a dynamic scale {S, 1 / S}, a
    row limit, in one launch with float32 jobs, and bit-identical when repeated."""
    rs = np.random.RandomState(P % 977)
    # adjoint-like operands pre-scaled into fp16's range (what nrh_adjoint_range arranges), activation-like ones as they are
    A1 = cu(_wide_range(rs, P, 256, 1.0)).clamp_(-6e4, 6e4)
    A2 = cu(rs.randn(P, 256).astype(np.float32) * 0.2)
    B1 = cu(np.log1p(np.exp(rs.randn(P, 256).astype(np.float32) * 3)) * 0.1)
    B2 = cu(_wide_range(rs, P, 256, 0.1)).clamp_(-6e4, 6e4)
    F1, G1 = cu(rs.randn(P, 256).astype(np.float32)), cu(rs.randn(P, 256).astype(np.float32))
    h = dw.to_half_tiled
    A1h, A2h, B1h, B2h = h(A1), h(A2), h(B1), h(B2)
    assert torch.equal(dw.from_half_tiled(A1h), A1.half())
    new = lambda *s: torch.full(s, float("nan"), dtype=torch.float32, device="cuda")
    out = dict(two=new(256, 256), btwo=new(256), one=new(217, 256), bone=new(217), f32=new(256, 256))
    dyn = torch.tensor([64.0, 1.0 / 64.0], device="cuda")
    jobs = [dw.Job([A1h, A2h], [B1h, B2h], 256, 256, out["two"], colsum_a=out["btwo"], half=True, dyn_scale=dyn),
            dw.Job([F1], [G1], 256, 256, out["f32"]),
            dw.Job([A2h], [B1h], 256, 256, out["one"], rows=217, scale=2.0 ** -0.5, colsum_a=out["bone"], half=True)]
    dw.run(jobs, P)
    torch.cuda.synchronize()
    d = lambda t: t.double().cpu()
    a1, a2, b1, b2 = (d(x.half()) for x in (A1, A2, B1, B2))

    def check(got, ref, absref, what, rel=2e-6):
        assert bool(torch.isfinite(d(got)).all()), what
        err = (d(got) - ref).abs()
        assert float((err / (absref + 1e-300)).max()) < rel, (what, float((err / (absref + 1e-300)).max()))

    check(out["two"], (a1.t() @ b1 + a2.t() @ b2) / 64, (a1.abs().t() @ b1.abs() + a2.abs().t() @ b2.abs()) / 64, "two pairs, dynamic scale")
    check(out["btwo"], a1.sum(0) / 64, a1.abs().sum(0) / 64, "column sums of A_0")
    check(out["one"], (a2.t() @ b1)[:217] * 2.0 ** -0.5, (a2.abs().t() @ b1.abs())[:217], "one pair, 217 rows")
    check(out["bone"], a2.sum(0)[:217], a2.abs().sum(0)[:217], "column sums, row limit")
    check(out["f32"], d(F1).t() @ d(G1), d(F1).abs().t() @ d(G1).abs(), "float32 job of the same launch", rel=3e-5)
    first = {k: v.clone() for k, v in out.items()}
    dw.run(jobs, P)
    for k in out:
        assert torch.equal(first[k], out[k]), k


def test_embedding_rows_vs_oracle():
    lib = _lib.load()
    n, npr = 37, 128
    o, dd, pl, near, far = make_rays(n, seed=3, spread=0.1)
    t = torch.rand(n, npr) * 2 + 2
    rows = torch.empty(n * npr, 64, device="cuda")
    P = _lib.ptr
    o_, d_, t_ = cu(o), cu(dd), t.cuda()           # (kept alive: a temporary's memory would be handed to the next allocation)
    _lib.check(lib.nrh_embedding_rows(P(o_), P(d_), P(t_), npr, npr, n, P(rows), _lib.stream_handle()), "emb")
    torch.cuda.synchronize()
    pts = (T(o)[:, None] + T(dd)[:, None] * t[..., None]).reshape(-1, 3)
    want = orc.nerf_encode(pts.double() * 3.0, 6)
    # sine of arguments up to 3 * 32 * |p| ~ 300: the float32 argument itself carries 300 * 2^-24 = 2e-5 of rounding
    assert float((rows[:, :39].cpu().double() - want).abs().max()) < 3e-5
    assert float((rows[:, :3].cpu().double() - want[:, :3]).abs().max()) < 2e-6      # the raw coordinates
    assert float(rows[:, 39:].abs().max()) == 0.0


@pytest.mark.parametrize("bg_on", [True, False])
def test_composite_loss_kernels_vs_autograd(bg_on):
    """nrh_composite_loss + nrh_loss_finish against torch autograd (float64) of the reference's expressions:
    rgb (:635-637), L1 colour loss / eikonal loss / psnr (pipelines/base_pipeline.py:57-69) and d loss / d (colour logits, weights)."""
    lib = _lib.load()
    rs = np.random.RandomState(2)
    n, igr, inv_s = 203, 0.1, 437.0
    col = rs.rand(n * 128, 3)
    w = rs.rand(n, 128) * 0.02
    gt = rs.rand(n, 3)
    gt[3] = 0.25                                          # (an exact hit of the L1 kink is exercised through rgb == gt below)
    nrm = rs.randn(n * 128, 3) * 0.8
    ins = (rs.rand(n, 128) < 0.7).astype(np.float64)
    bg = np.array([1.0, 0.5, 0.0])
    t64 = lambda a, g=False: torch.tensor(a, dtype=torch.float64, requires_grad=g)
    c_, w_, g_ = t64(col, True), t64(w, True), t64(nrm, True)
    rgb = (c_.reshape(n, 128, 3) * w_[..., None]).sum(1)
    if bg_on:
        rgb = rgb + t64(bg) * (1.0 - w_.sum(-1, keepdim=True))
    rgb_loss = (rgb - t64(gt)).abs().sum() / (n + 1e-5)
    eik = (t64(ins) * (torch.linalg.norm(g_.reshape(n, 128, 3), dim=-1) - 1.0) ** 2).sum() / (t64(ins).sum() + 1e-5)
    loss = rgb_loss + igr * eik
    cb, wb, gb = torch.autograd.grad(loss, [c_, w_, g_])
    f = lambda a: cu(np.asarray(a, dtype=np.float32))
    new = lambda *s: torch.empty(*s, device="cuda")
    o_rgb, zbar4, wbar, part, loss8 = new(n, 3), new(n * 128, 3), new(n, 128), new(n, 4), new(8)
    P = _lib.ptr
    col_, w_g, gt_, bg_, nrm_, ins_ = f(col), f(w), f(gt), f(bg), f(nrm), f(ins)      # kept alive across the calls
    _lib.check(lib.nrh_composite_loss(P(col_), P(w_g), P(gt_), P(bg_) if bg_on else None, P(nrm_), P(ins_), n, P(o_rgb), P(zbar4),
                                      P(wbar), P(part), _lib.stream_handle()), "composite_loss")
    _lib.check(lib.nrh_loss_finish(P(part), n, inv_s, None, igr, P(loss8), _lib.stream_handle()), "loss_finish")
    np.testing.assert_allclose(o_rgb.cpu().numpy(), rgb.detach().numpy(), rtol=0, atol=2e-6)
    got = loss8.cpu().double().numpy()
    np.testing.assert_allclose(got[0], loss.item(), rtol=2e-6)
    np.testing.assert_allclose(got[1], rgb_loss.item(), rtol=2e-6)
    np.testing.assert_allclose(got[2], eik.item(), rtol=2e-6)
    np.testing.assert_allclose(got[3], 1.0 / inv_s, rtol=1e-6)
    np.testing.assert_allclose(got[4], (10.0 * torch.log10(1.0 / ((rgb - t64(gt)) ** 2).mean())).item(), rtol=1e-5)
    np.testing.assert_allclose(got[5], igr / (ins.sum() + 1e-5), rtol=1e-6)
    # seeds: zbar4 = d loss / d (pre-sigmoid colour) = cbar c (1 - c); wbar = d loss / d w
    np.testing.assert_allclose(zbar4.cpu().numpy(), (cb * c_.detach() * (1 - c_.detach())).numpy(), rtol=1e-5, atol=1e-10)   # values ~1e-5
    np.testing.assert_allclose(wbar.cpu().numpy(), wb.numpy(), rtol=1e-4, atol=1e-9)
    # the eikonal seed through the extended alpha adjoint: everything else zero
    sdf = f(rs.uniform(-0.02, 0.05, size=(n, 128)))
    dirs = rs.randn(n, 3); dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    dists = f(rs.uniform(0.002, 0.03, size=(n, 128)))
    sb, gbar, rdb, ib = new(n * 128), new(n * 128, 3), new(n, 3), new(n)
    zeros = torch.zeros(n, 128, device="cuda")
    import ctypes
    dirs_ = f(dirs)
    _lib.check(lib.nrh_alpha_train_backward_fused(P(sdf), P(nrm_), P(dirs_), P(dists), inv_s, 1.0, None, n, P(zeros), None, 3, P(ins_),
                                                  ctypes.c_void_p(loss8.data_ptr() + 20), P(sb), P(gbar), P(rdb), P(ib), 128, _lib.stream_handle()),
               "alpha_fused")
    np.testing.assert_allclose(gbar.cpu().numpy(), gb.numpy(), rtol=2e-5, atol=1e-10)
    assert float(sb.abs().max()) == 0.0


def test_alpha_adjoint_strided_normal_bar():
    """nhat_bar given as columns 3..5 of a [P,128] array (the reflectance adjoint's output) == the contiguous copy."""
    lib = _lib.load()
    rs = np.random.RandomState(4)
    n = 19
    f = lambda a: cu(np.asarray(a, dtype=np.float32))
    sdf, grad = f(rs.uniform(-0.02, 0.05, size=(n, 128))), f(rs.randn(n * 128, 3) * 0.7)
    dirs = rs.randn(n, 3); dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    dists, wb = f(rs.uniform(0.002, 0.03, size=(n, 128))), f(rs.randn(n, 128))
    mbar = f(rs.randn(n * 128, 128))
    nb = mbar[:, 3:6].contiguous()
    new = lambda *s: torch.empty(*s, device="cuda")
    P = _lib.ptr
    import ctypes
    outs = []
    dirs_ = f(dirs)
    for ptr, stride in ((P(nb), 3), (ctypes.c_void_p(mbar.data_ptr() + 12), 128)):
        o = [new(n * 128), new(n * 128, 3), new(n, 3), new(n)]
        _lib.check(lib.nrh_alpha_train_backward_fused(P(sdf), P(grad), P(dirs_), P(dists), 300.0, 0.6, None, n, P(wb), ptr, stride, None, None,
                                                      *(P(x) for x in o), 128, _lib.stream_handle()), "alpha_fused")
        outs.append(o)
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("prec", ["f16x3", "f32"])
def test_fused_step_gradients_vs_reference(scene_states, tag, prec):
    """train_fused.train_step_backward against what the imported reference produced for the same batch and jitter
    (tests/golden/train_*.npz): loss dict, and the gradient of every parameter tensor against the reference's float64 gradient within
    the bound derived from the reference's own float32 noise (tests/conftest.py grad_bound)."""
    g = load_npz(f"train_{tag}.npz")
    model = _model(scene_states[tag], prec)
    rb = _bundle(*(g[k] for k in ("o", "d", "pl", "near", "far")))
    assert train_fused.supported(model, rb) is None
    loss8 = train_fused.train_step_backward(model, rb, cu(g["rgb_gt"]), torch.ones(1, 3).cuda(), int(g["global_step"]),
                                            t_rand_primary=cu(g["t_rand_primary"]), t_rand_shadow=cu(g["t_rand_shadow"]))
    ld = train_fused.loss_dict(loss8)
    np.testing.assert_allclose(ld["loss"], g["loss"], rtol=2e-4)
    np.testing.assert_allclose(ld["rgb_loss"], g["rgb_loss"], rtol=2e-4)
    np.testing.assert_allclose(ld["eikonal_loss"], g["eikonal_loss"], rtol=2e-3)
    report = []
    for name, prm in model.named_parameters():
        assert prm.grad is not None and prm.grad.shape == prm.shape, name
        tol, scale = grad_bound(g["grad." + name], g["grad64." + name])
        err = float(np.abs(prm.grad.detach().cpu().numpy().astype(np.float64) - g["grad64." + name]).max())
        report.append((err / tol, name, err / scale, tol / scale))
    bad = [r for r in report if r[0] >= 1.0]
    assert not bad, "gradient outside its derived bound (ratio, tensor, err/scale, bound/scale): " + repr(sorted(bad, reverse=True)[:8])


@pytest.mark.parametrize("tag,prec", [("a", "f16x3"), ("b", "f16x3"), ("b", "f32")])
def test_fused_step_ray_gradients_vs_reference(scene_states, tag, prec):
    """Pose / light refinement on the fused step (nr-hints-cam-opt): d loss / d (origins, directions,
    pl_positions) from nrh_ray_adjoint against the reference's recorded float64 gradients of the same training step
    (tests/golden/train_*.npz), bounds derived from its own float32 run - the same check the autograd path passes in
    test_gpu_parity.py::test_training_step_gradients - and the parameter gradients unchanged by asking for them."""
    g = load_npz(f"train_{tag}.npz")
    model = _model(scene_states[tag], prec)
    rb = _bundle(*(g[k] for k in ("o", "d", "pl", "near", "far")))
    for t_ in (rb.origins, rb.directions, rb.pl_positions):
        t_.requires_grad_(True)
    assert train_fused.supported(model, rb) is None
    grads = {}
    loss8 = train_fused.train_step_backward(model, rb, cu(g["rgb_gt"]), torch.ones(1, 3).cuda(), int(g["global_step"]),
                                            t_rand_primary=cu(g["t_rand_primary"]), t_rand_shadow=cu(g["t_rand_shadow"]), ray_grads=grads)
    np.testing.assert_allclose(float(loss8[0]), float(g["loss"]), rtol=2e-4)
    bad = []
    for nm in ("origins", "directions", "pl_positions"):
        tol, scale = grad_bound(g["grad.rays." + nm], g["grad64.rays." + nm])
        err = float(np.abs(grads[nm].cpu().numpy().astype(np.float64) - g["grad64.rays." + nm]).max())
        if err >= tol:
            bad.append((nm, err / scale, tol / scale))
    assert not bad, bad
    assert rb.origins.grad is None                      # returned, not propagated
    for name in ("sdf_network.lin4.weight_v", "color_network.lin0.weight_v", "deviation_network.variance"):
        tol, scale = grad_bound(g["grad." + name], g["grad64." + name])
        prm = dict(model.named_parameters())[name]
        assert float(np.abs(prm.grad.detach().cpu().numpy().astype(np.float64) - g["grad64." + name]).max()) < tol, name
    # without a dict the adjoints go into the graph behind the bundle: leaves receive .grad, as loss.backward() would give
    train_fused.train_step_backward(model, rb, cu(g["rgb_gt"]), torch.ones(1, 3).cuda(), int(g["global_step"]),
                                    t_rand_primary=cu(g["t_rand_primary"]), t_rand_shadow=cu(g["t_rand_shadow"]))
    for nm, t_ in (("origins", rb.origins), ("directions", rb.directions), ("pl_positions", rb.pl_positions)):
        assert torch.equal(t_.grad, grads[nm]), nm


def test_fused_register_view_step_equals_autograd(scene_states):
    """The step register_view takes (pipelines/base_pipeline.py:80-91): evaluation-mode forward, L1 / (N + 1e-5), gradients for
    the rays only.  Fused with the renderer frozen (no weight-gradient launches) against the autograd path on the same rays."""
    model = _model(scene_states["b"]).eval()
    n = 256
    rs = np.random.RandomState(9)
    bg = torch.ones(1, 3).cuda()
    gt = cu(rs.rand(n, 3).astype(np.float32))
    rb = _bundle(*make_rays(n, seed=21, spread=0.1))
    for t_ in (rb.origins, rb.directions, rb.pl_positions):
        t_.requires_grad_(True)
    out = model(rb, background_rgb=bg, is_training=False)
    loss = torch.nn.functional.l1_loss(out.rgb, gt, reduction="sum") / (n + 1e-5)
    loss.backward()
    want = [t_.grad.clone() for t_ in (rb.origins, rb.directions, rb.pl_positions)]
    model.zero_grad(set_to_none=True)
    for p in model.parameters():
        p.requires_grad_(False)
    grads = {}
    loss8 = train_fused.train_step_backward(model, rb, gt, bg, 0, igr_weight=0.0, is_training=False, ray_grads=grads)
    np.testing.assert_allclose(float(loss8[0]), float(loss.detach()), rtol=1e-5)
    np.testing.assert_allclose(float(loss8[1]), float(loss.detach()), rtol=1e-5)
    for nm, w in zip(("origins", "directions", "pl_positions"), want):
        scale = float(w.abs().max()) + 1e-30
        assert float((grads[nm] - w).abs().max()) < 1e-4 * scale + 1e-7, (nm, float((grads[nm] - w).abs().max()), scale)
    assert all(p.grad is None for p in model.parameters())


@pytest.mark.parametrize("half", [False, True])
def test_fused_step_equals_autograd_path(scene_states, half):
    """Same batch, same jitter through the autograd Functions (forward + train_loss_dict + backward) and through the fused
    sequence: same kernels for the sweeps and the weight gradients, so the results agree to fp32 round-off of the few
    elementwise expressions that moved from torch into the composite / loss kernels.  ``half``: the fused step with the OPTIONAL
    16-bit hand-offs of the SDF net's weight-gradient operands (renderer.dw_half; the autograd path keeps float32 arrays and the
    bf16x3 products): the operands' 11-bit rounding shows as up to 1.5e-4 of a tensor's scale - the price measured against the
    reference in tests/test_gpu_train1024.py and profiles/r05/dw16_emulation.log."""
    from nrhints_amd.training import train_loss_dict
    n = 256
    rs = np.random.RandomState(3)
    rb = _bundle(*make_rays(n, seed=12, spread=0.1))
    gt, tp, ts = (cu(rs.rand(n, k).astype(np.float32)) for k in (3, 1, 64))
    bg = torch.ones(1, 3).cuda()
    a, b = _model(scene_states["b"]), _model(scene_states["b"])
    b.dw_half = half
    out = a(rb, is_training=True, background_rgb=bg, global_step=30000, _t_rand_primary=tp, _t_rand_shadow=ts)
    la = train_loss_dict(out, gt, a.config.igr_weight)
    la["loss"].backward()
    loss8 = train_fused.train_step_backward(b, rb, gt, bg, 30000, t_rand_primary=tp, t_rand_shadow=ts)
    lb = train_fused.loss_dict(loss8)
    for k in ("loss", "rgb_loss", "eikonal_loss", "s_val", "psnr"):
        np.testing.assert_allclose(lb[k], float(la[k]), rtol=5e-6, err_msg=k)
    for (name, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        scale = float(pa.grad.abs().max()) + 1e-30
        # 1e-4 of the tensor's scale, plus 1e-6 absolute: the bias gradients of the last reflectance layers are sums over 32 768
        # samples that cancel to ~1e-3, so the fp32 round-off of the summands (5e-7 absolute between the two paths) is not small
        # against the RESULT although it is against every term
        err = float((pa.grad - pb.grad).abs().max())
        assert err < (5e-4 if half else 1e-4) * scale + 1e-6, (name, err, scale)


def test_fused_training_descends_and_graph_replays(scene_states):
    """Eight fused optimisation steps on a fixed batch bring the loss down, every parameter moves; and a GraphedTrainStep over the
    fused body replays bit-identically to eager fused steps with the same optimiser arithmetic."""
    from nrhints_amd.training import GraphedTrainStep, lr_factor, make_optimizer, train_step
    from nrhints_amd.synthetic import perturb_state
    torch.manual_seed(0)
    student = na.NeuSHintRenderer().cuda()
    teacher = _model(scene_states["b"]).eval()
    rb = _bundle(*make_rays(256, seed=5, spread=0.08))
    bg = torch.ones(1, 3).cuda()
    with torch.no_grad():
        gt = teacher(rb, background_rgb=bg).rgb
    before = {k: v.detach().clone() for k, v in student.named_parameters()}
    opt, sched = make_optimizer(student, lr=1e-3, warm_up_end=1)
    torch.manual_seed(1)
    losses = [train_step(student, rb, gt, bg, 50_000, opt, sched, fused=True)["loss"] for _ in range(8)]
    assert all(np.isfinite(losses)) and losses[-1] < 0.8 * losses[1], losses
    assert sum(not torch.equal(v.detach(), before[k]) for k, v in student.named_parameters()) == 46
    # graph replay == eager, both fused
    n, lr, gs = 128, 5e-4, 30000
    rs = np.random.RandomState(5)
    batches = [(_bundle(*make_rays(n, seed=40 + i, spread=0.1)), cu(rs.rand(n, 3).astype(np.float32))) for i in range(3)]
    tp, ts = cu(rs.rand(n, 1).astype(np.float32)), cu(rs.rand(n, 64).astype(np.float32))
    eager, graphed = _model(scene_states["b"]), _model(scene_states["b"])
    lr_t = torch.tensor(lr, device="cuda")
    from nrhints_amd.adam import HipAdam
    opt = HipAdam([{"params": list(eager.parameters()), "lr": lr_t}])        # the optimiser GraphedTrainStep uses around the fused body
    step = GraphedTrainStep(graphed, n, bg, lr=lr, warm_up_end=20, global_step=gs, jitter=(tp, ts), fused=True)
    for i, (rb_i, gt_i) in enumerate(batches):
        lr_t.fill_(lr * lr_factor(gs + i, 20, 1_000_000, 0.05))
        opt.zero_grad(set_to_none=True)
        l8 = train_fused.train_step_backward(eager, rb_i, gt_i, bg, gs + i, t_rand_primary=tp, t_rand_shadow=ts)
        want = train_fused.loss_dict(l8)
        opt.step()
        got = step(rb_i, gt_i, global_step=gs + i)
        assert got["loss"] == want["loss"], (i, got["loss"], want["loss"])
        for (k, pe), (_, pg) in zip(eager.named_parameters(), graphed.named_parameters()):
            assert torch.equal(pe.detach(), pg.detach()), (i, k)
    step.release()


def test_pack_plans_on_gpu_bit_identical(scene_states):
    """nrh_pack_gather / nrh_sdf32_tables (the re-pack of a training step as three launches) against the torch form of the same
    index plans on the CPU: every packed buffer bit for bit (same conversions, same IEEE division)."""
    from nrhints_amd import packing as pk, packing32 as pk32
    st = {k: T(np.asarray(v)) for k, v in scene_states["b"].items()}
    d_cpu = pk.dense_params(st)
    d_gpu = {k: v.cuda() for k, v in d_cpu.items()}
    for prec in (0, 1):
        want = pk.PackPlan(d_cpu, prec, True).pack(d_cpu)
        got = pk.PackPlan(d_gpu, prec, True).pack(d_gpu)
        for k in want:
            a, b = want[k], got[k].cpu()
            assert a.dtype == b.dtype and torch.equal(a.view(torch.int16 if a.dtype == torch.float16 else torch.int32),
                                                      b.view(torch.int16 if b.dtype == torch.float16 else torch.int32)), (prec, k)
        # ... and the DIRECT packers run on GPU tensors (what an evaluation render packs with) give the same bits as the plan (what a
        # training step packs with): round 6 found W4 / sqrt(2) a multiplication by the reciprocal there (torch's tensor / scalar on
        # the GPU) - a last-bit difference that moved importance samples between the two routes
        sw, sb, sh = pk.pack_sdf(d_gpu, prec)
        cw, cb = pk.pack_color(d_gpu, prec, True)
        for name, direct in (("sdf_w", sw), ("sdf_b", sb), ("sdf_head", sh), ("col_w", cw), ("col_b", cb),
                             ("sdf_wt_feat", pk.pack_feat_transposed(d_gpu, prec))):
            a, b = got[name], direct
            assert a.dtype == b.dtype and torch.equal(a.view(torch.int16 if a.dtype == torch.float16 else torch.int32),
                                                      b.view(torch.int16 if b.dtype == torch.float16 else torch.int32)), (prec, name, "direct packer")
    ws, wt = pk32.PackPlan32(d_cpu).pack(d_cpu)
    gs, gt = pk32.PackPlan32(d_gpu).pack(d_gpu)
    assert torch.equal(ws.view(torch.int16), gs.cpu().view(torch.int16))
    assert torch.equal(wt.view(torch.int32), gt.cpu().view(torch.int32))


def test_hip_adam_equals_torch_adam():
    """adam.HipAdam (nrh_adam_step: one launch for all tensors) against torch.optim.Adam - the default implementation the
    reference trains with - on the same parameters and gradients over several steps, two parameter groups with different
    learning rates (a device tensor on the HIP side: graph replays read it at run time), INCLUDING lr = 0 with all-zero gradient
    entries (step 0 of the warm-up schedule; torch's capturable variant returns NaN there): parameters / moments equal to a few
    ulp, same state keys, and the state dict loads into a plain torch Adam."""
    from nrhints_amd.adam import HipAdam
    rs = np.random.RandomState(0)
    shapes = [(256, 39), (256, 1), (256,), (), (3, 256), (217, 256), (5000,)]
    mk = lambda r: [torch.nn.Parameter(cu(r.randn(*s).astype(np.float32) if s else np.array(r.randn(), dtype=np.float32))) for s in shapes]
    pa, pb = mk(np.random.RandomState(1)), mk(np.random.RandomState(1))
    lr_t = torch.tensor(0.0, device="cuda")
    hip = HipAdam([{"params": pa[:5], "lr": lr_t}, {"params": pa[5:], "lr": 1e-3}])
    ref = torch.optim.Adam([{"params": pb[:5], "lr": 0.0}, {"params": pb[5:], "lr": 1e-3}])
    grads = [torch.empty_like(p) for p in pa]
    for p, g in zip(pa, grads):
        p.grad = g
    for it in range(5):
        lr0 = 5e-4 * it / 5                                   # 0 on the first step
        lr_t.fill_(lr0)
        ref.param_groups[0]["lr"] = lr0
        for g, q in zip(grads, pb):
            g.copy_(cu(rs.randn(*g.shape).astype(np.float32) if g.dim() else np.array(rs.randn(), dtype=np.float32)) * (10.0 ** rs.randint(-6, 1)))
            if g.dim() == 2:
                g[:3].zero_()                                 # entries that never see a gradient
            q.grad = g.clone()
        hip.step(); ref.step()
        for i, (a, b) in enumerate(zip(pa, pb)):
            assert bool(torch.isfinite(a).all()), (it, i)
            tol = 4e-7 * float(b.detach().abs().max()) + 1e-9
            assert float((a.detach() - b.detach()).abs().max()) <= tol, (it, i)
    assert hip._stage is None                       # fixed gradient addresses: no staging copy
    sa, sb = hip.state_dict(), ref.state_dict()
    assert sa["state"].keys() == sb["state"].keys() and len(sa["param_groups"]) == 2
    for k in sa["state"]:
        assert float(sa["state"][k]["step"]) == 5.0 == float(sb["state"][k]["step"])
        for name in ("exp_avg", "exp_avg_sq"):
            x, y = sa["state"][k][name], sb["state"][k][name]
            assert float((x - y).abs().max()) <= 4e-7 * float(y.abs().max()) + 1e-30, (k, name)
    torch.optim.Adam([{"params": pb[:5]}, {"params": pb[5:]}]).load_state_dict(sa)
    # gradients that move every step: staged through the optimiser's own buffers, results unchanged
    hip2, ref2 = HipAdam([{"params": pa, "lr": 1e-3}]), torch.optim.Adam([{"params": pb, "lr": 1e-3}])
    with torch.no_grad():
        for a, b in zip(pa, pb):
            b.copy_(a)
    for it in range(4):
        for a, b in zip(pa, pb):
            a.grad = torch.randn_like(a)
            b.grad = a.grad.clone()
        hip2.step(); ref2.step()
    assert hip2._stage is not None
    for a, b in zip(pa, pb):
        assert float((a.detach() - b.detach()).abs().max()) <= 4e-7 * float(b.detach().abs().max()) + 1e-9


def test_train_step_without_readback_equals_with(scene_states):
    """training.train_step(sync=False) - no per-step loss read-back, 1/s and the cos-anneal ratio on the device - takes the same
    steps as the default call: same losses, same parameters after four steps (fixed jitter through the seed), and a step WITH the
    read-back afterwards still follows the anneal schedule (the kernels read the device scalars once they exist)."""
    from nrhints_amd.training import make_optimizer, release_device_scalars, train_step
    rb = _bundle(*make_rays(128, seed=9, spread=0.08))
    bg = torch.ones(1, 3).cuda()
    gt = cu(np.random.RandomState(3).rand(128, 3).astype(np.float32))
    runs = []
    for sync in (True, False):
        model = _model(scene_states["b"])
        opt, sched = make_optimizer(model, lr=5e-4, warm_up_end=2)
        torch.manual_seed(11)
        losses = [train_step(model, rb, gt, bg, 10_000 + 5_000 * i, opt, sched, sync=sync)["loss"] for i in range(4)]
        if not sync:
            assert model.dyn_scalars is not None and all(torch.is_tensor(x) for x in losses)
            assert abs(float(model.dyn_scalars[1]) - 25_000 / 50_000) < 1e-7
            more = train_step(model, rb, gt, bg, 40_000, opt, sched, sync=True)["loss"]       # read-back variant on device scalars
            assert abs(float(model.dyn_scalars[1]) - 0.8) < 1e-7 and np.isfinite(more)
            release_device_scalars(model)
            assert model.dyn_scalars is None
            with torch.no_grad():
                assert bool(torch.isfinite(model(rb, background_rgb=bg).rgb).all())
        runs.append(([float(x) for x in losses], {k: v.detach().clone() for k, v in model.named_parameters()}))
    (la, pa), (lb, pb) = runs
    np.testing.assert_allclose(la, lb, rtol=1e-4)
    # (pb took one more step; compare what both did: the losses above, and that the fifth step moved the parameters only slightly)
    for k in pa:
        assert float((pa[k] - pb[k]).abs().max()) < 2e-3, k


def _branch_case(vt, scene_states):
    """(renderer config, state dict, fixture, key prefix) of the off-default branches the fused step covers since round 5"""
    from nrhints_amd.synthetic import one_hint_state
    R, sb = na.NeuSRendererConfig, scene_states["b"]
    if vt == "ana":
        return R(normal_type=na.NormalComputationType.Analytic), sb, load_npz("train_analytic_b.npz")
    g = load_npz("render_branches_b.npz")
    if vt == "sho":
        return R(shadow_hint=True, specular_hint=False), one_hint_state(sb, True), g
    if vt == "spo":
        return R(shadow_hint=False, specular_hint=True), one_hint_state(sb, False), g
    return R(n_shadow_importance_clip=8), sb, g


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
@pytest.mark.parametrize("vt", ["sho", "spo", "psh", "ana"])
def test_fused_step_off_default_branches(scene_states, vt, prec):
    """The autograd-free step for the branches that differ from the default model only in what the alpha / colour stages are fed
    (VERDICT r4 item 5): one hint without the other (models/neus_hint_model.py:246-257), the partial visibility hint (:554-576) and
    Analytic normals (:622-623).  Against the reference's recorded training step (loss; the recorded gradient tensors against its
    float64 run within the bounds the autograd path's tests of the same fixtures use), against the autograd path on the same batch
    (same kernels: float32 round-off), and as a captured hipGraph replay (learning rate 0: bit-equal loss, parameters untouched)."""
    from nrhints_amd.training import GraphedTrainStep, train_loss_dict
    rcfg, st, g = _branch_case(vt, scene_states)
    cfg = na.NeuSModelConfig(renderer=rcfg)
    gs = int(g["t.global_step"])
    tb = lambda: _bundle(*(g["t." + k] for k in ("o", "d", "pl", "near", "far")))
    tp = cu(g[f"{vt}.t_rand_primary"])
    ts = cu(g[f"{vt}.t_rand_shadow"]) if vt != "spo" else None
    gt, bg = cu(g["t.rgb_gt"]), torch.ones(1, 3).cuda()
    fused, auto = _model(st, prec, cfg).train(), _model(st, prec, cfg).train()
    rb = tb()
    assert train_fused.supported(fused, rb) is None
    loss8 = train_fused.train_step_backward(fused, rb, gt, bg, gs, t_rand_primary=tp, t_rand_shadow=ts)
    ld = train_fused.loss_dict(loss8)
    np.testing.assert_allclose(ld["loss"], float(g[f"{vt}.loss"]), rtol=2e-4)
    # the autograd path on the same batch and jitter
    out = auto(tb(), is_training=True, background_rgb=bg, global_step=gs, _t_rand_primary=tp, _t_rand_shadow=ts)
    la = train_loss_dict(out, gt, auto.config.igr_weight)
    la["loss"].backward()
    np.testing.assert_allclose(ld["loss"], float(la["loss"]), rtol=5e-6)
    for (name, pa), (_, pf) in zip(auto.named_parameters(), fused.named_parameters()):
        assert pf.grad is not None and pf.grad.shape == pf.shape == pa.grad.shape, name
        scale = float(pa.grad.abs().max()) + 1e-30
        err = float((pa.grad - pf.grad).abs().max())
        # (5e-6 absolute: out_sdf.bias is a sum of 4 096 adjoints of either sign that cancels to 4e-4 on these 32-ray batches)
        assert err < 1e-4 * scale + 5e-6, (vt, name, err, scale)
    # the reference's recorded tensors (psh: its group visibilities sit on the surface, where single flips move the gradients by
    # per cents - the autograd path's own test of this fixture checks direction + magnitude only; the equality above carries it here)
    if vt != "psh":
        named = dict(fused.named_parameters())
        keys = [k for k in g if k.startswith(f"{vt}.grad.") and ".rays." not in k]
        assert len(keys) == 11
        for k in keys:
            want64 = g[k.replace(".grad.", ".grad64.")]
            bound, scale = grad_bound(g[k], want64, factor=4.0, floor=5e-3)      # 32 rays: one coarse draw of the reference's own noise
            err = float(np.abs(named[k[len(vt) + 6:]].grad.detach().cpu().numpy().astype(np.float64) - want64).max())
            assert err <= bound, (vt, k, err, bound, scale)
    # captured: GraphedTrainStep takes the fused body for these branches now
    graphed = _model(st, prec, cfg).train()
    before = {k: v.detach().clone() for k, v in graphed.named_parameters()}
    jit = (tp.reshape(-1, 1), ts if ts is not None else torch.zeros(tp.numel(), 64, device="cuda"))
    step = GraphedTrainStep(graphed, tp.numel(), bg, lr=0.0, warm_up_end=0, global_step=gs, jitter=jit)
    try:
        assert step._use_fused
        got = step(tb(), gt, global_step=gs)
        assert got["loss"] == ld["loss"], (got["loss"], ld["loss"])
        for (name, pg), (_, pf) in zip(graphed.named_parameters(), fused.named_parameters()):
            assert torch.equal(pg.grad, pf.grad), name
            assert torch.equal(pg.detach(), before[name]), name
    finally:
        step.release()


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
def test_every_path_places_the_same_samples(scene_states, prec):
    """The same parameters, rays and jitter place the SAME samples whichever path evaluates them - the fused step's forward
    (train_step_backward forward_out), the autograd path's forward, a bare _render_train and the evaluation pack - bit for bit.
    Round 6 found them different: the evaluation pack folded weight-norm with torch ops, the training paths with
    nrh_weight_norm_fold; the two W differ in last bits, and a 1e-7 change of the SDF moves importance samples by a whole bin
    where the pdf sits at its floor (20 % of the samples of a 1 024-ray batch).  One fold kernel for all of them now
    (renderer.packed_params)."""
    n = 512
    rs = np.random.RandomState(11)
    rb = _bundle(*make_rays(n, seed=21, spread=0.1))
    gt, tp, ts = (cu(rs.rand(n, k).astype(np.float32)) for k in (3, 1, 64))
    bg = torch.ones(1, 3).cuda()
    f32 = lambda t: t.detach().float().contiguous()
    # 1. a bare training forward on a fresh model (pack built by packed_params(dense=None), the evaluation pack's route)
    a = _model(scene_states["b"], prec)
    res = a._render_train(f32(rb.origins), f32(rb.directions), f32(rb.pl_positions), f32(rb.nears).reshape(-1), f32(rb.fars).reshape(-1),
                          30000 / a.config.anneal_end, tp.reshape(-1).contiguous(), ts, 0)
    # 2. the fused step on another fresh model
    b = _model(scene_states["b"], prec)
    fwd = {}
    train_fused.train_step_backward(b, rb, gt, bg, 30000, t_rand_primary=tp, t_rand_shadow=ts, forward_out=fwd)
    # 3. the autograd path on a third
    c = _model(scene_states["b"], prec)
    out = c(rb, is_training=True, background_rgb=bg, global_step=30000, _t_rand_primary=tp, _t_rand_shadow=ts)
    for k in ("mid_z", "dists", "weights", "visibilities", "depth"):
        assert torch.equal(res[k], fwd[k]), (k, float((res[k] - fwd[k]).abs().max()))
    assert torch.equal(res["pre"]["sdf"], fwd["sdf"]) and torch.equal(res["normals"], fwd["normals"])
    assert torch.equal(out.weights.detach(), fwd["weights"]) and torch.equal(out.visibilities.detach(), fwd["visibilities"])
    assert torch.equal(out.analytic_normals.detach(), fwd["normals"])
    # ... and after the fused step the same model's bare forward still places them there (the cached pack is the step's)
    res2 = b._render_train(f32(rb.origins), f32(rb.directions), f32(rb.pl_positions), f32(rb.nears).reshape(-1), f32(rb.fars).reshape(-1),
                           30000 / b.config.anneal_end, tp.reshape(-1).contiguous(), ts, 0)
    assert torch.equal(res2["mid_z"], fwd["mid_z"]) and torch.equal(res2["weights"], fwd["weights"])


def test_step_scalars_and_alpha_stage_points(scene_states):
    """ABI 148's launch savers: nrh_step_scalars writes up to four host floats and 1 / s = clip(exp(10 variance), 1e-6, 1e6) in one
    launch (torch's pointwise kernels before); the alpha stage leaves the reflectance net's point input p = o + d * mid_z with the
    roundings of the torch expression it replaced (fl(fl(d t) + o), what the SDF kernels form)."""
    a, b, c = torch.zeros(2, device="cuda"), torch.zeros((), device="cuda"), torch.zeros((), device="cuda")
    _lib.step_scalars([(a[1:2], 0.37), (b, 5e-4), (c, -2.5)])
    assert a.tolist() == [0.0, float(np.float32(0.37))] and float(b) == float(np.float32(5e-4)) and float(c) == -2.5
    for v in (0.3, 0.7, -2.0, 2.0, float("nan")):
        var, out = torch.tensor([v], device="cuda"), torch.zeros(2, device="cuda")
        _lib.step_scalars(variance=var, inv_s_out=out)
        want = torch.exp(var * 10.0).clip(1e-6, 1e6)
        got = out[0:1]
        assert (torch.isnan(got).item() and torch.isnan(want).item()) or abs(float(got) - float(want)) <= 1.2e-7 * abs(float(want)), (v, float(got), float(want))
    assert float(out[1]) == 0.0
    # the points
    n = 96
    rs = np.random.RandomState(5)
    rb = _bundle(*make_rays(n, seed=13, spread=0.1))
    tp, ts = (cu(rs.rand(n, k).astype(np.float32)) for k in (1, 64))
    m = _model(scene_states["b"])
    f32 = lambda t: t.detach().float().contiguous()
    pts = torch.full((n * 128, 3), float("nan"), device="cuda")
    res = m._render_train(f32(rb.origins), f32(rb.directions), f32(rb.pl_positions), f32(rb.nears).reshape(-1), f32(rb.fars).reshape(-1), 0.6,
                          tp.reshape(-1).contiguous(), ts, 0, pts=pts)
    want = (rb.directions[:, None, :] * res["mid_z"][..., None])
    want = want + rb.origins[:, None, :]
    assert torch.equal(pts.view(n, 128, 3), want)


def test_adjoint_scale_from_seeds_for_sum_reduced_losses(scene_states):
    """_lib.ADJOINT_SCALE_FROM_SEEDS: the generic autograd Functions take their f16x3 adjoint scale from the incoming
    adjoints instead of the 1 / rays convention - for callers whose loss is not normalised by the ray count.  A loss multiplied by
    2^20 (adjoints of ~1e3 per sample: with the static scale rays / 8 = 32 on top, the SDF chain's largest seeds - inv_s / 4 per
    unit of colour adjoint - leave fp16's range) gives, with the option on, exactly 2^20 times the gradients of the plain loss
    (power-of-two scalings are exact through the chain); the default convention stays what the eager / graph identity tests pin."""
    from nrhints_amd.training import train_loss_dict
    n = 256
    rs = np.random.RandomState(4)
    rb = _bundle(*make_rays(n, seed=14, spread=0.1))
    gt, tp, ts = (cu(rs.rand(n, k).astype(np.float32)) for k in (3, 1, 64))
    bg = torch.ones(1, 3).cuda()

    def grads(scale, from_seeds):
        m = _model(scene_states["b"])
        old = _lib.ADJOINT_SCALE_FROM_SEEDS
        _lib.ADJOINT_SCALE_FROM_SEEDS = from_seeds
        try:
            out = m(rb, is_training=True, background_rgb=bg, global_step=30000, _t_rand_primary=tp, _t_rand_shadow=ts)
            (train_loss_dict(out, gt, m.config.igr_weight)["loss"] * scale).backward()
        finally:
            _lib.ADJOINT_SCALE_FROM_SEEDS = old
        return {k: v.grad.detach().clone() for k, v in m.named_parameters()}

    base = grads(1.0, False)
    big = grads(2.0 ** 20, True)
    for k in base:
        assert torch.isfinite(big[k]).all(), k
        want = base[k] * 2.0 ** 20
        scale = float(want.abs().max()) + 1e-30
        assert float((big[k] - want).abs().max()) <= 2e-4 * scale, (k, float((big[k] - want).abs().max()) / scale)


@pytest.mark.parametrize("vt", ["shg", "spg", "bhg"])
def test_fused_step_hint_gradients(scene_states, vt):
    """renderer.shadow_hint_gradient / specular_hint_gradient / both (models/neus_hint_model.py:379, :589) on the FUSED step (round 6;
    the autograd path had them since round 3): loss and the recorded gradient tensors against the reference's float64 run (the bounds
    of test_hint_gradients_vs_reference), every parameter gradient against the autograd path's, and the captured hipGraph takes the
    fused step and replays it to the eager numbers."""
    from nrhints_amd.training import GraphedTrainStep, train_loss_dict
    g = load_npz("render_branches_b.npz")
    R = na.NeuSRendererConfig
    cfg = na.NeuSModelConfig(renderer=R(shadow_hint_gradient=vt in ("shg", "bhg"), specular_hint_gradient=vt in ("spg", "bhg")))
    tb = _bundle(*(g["t." + k] for k in ("o", "d", "pl", "near", "far")))
    bg = torch.ones(1, 3).cuda()
    gs = int(g["t.global_step"])
    tp, ts, gt = cu(g[f"{vt}.t_rand_primary"]), cu(g[f"{vt}.t_rand_shadow"]), cu(g["t.rgb_gt"])
    fused = _model(scene_states["b"], cfg=cfg)
    assert train_fused.supported(fused, tb) is None
    loss8 = train_fused.train_step_backward(fused, tb, gt, bg, gs, t_rand_primary=tp, t_rand_shadow=ts)
    np.testing.assert_allclose(float(loss8[0]), float(g[f"{vt}.loss"]), rtol=2e-4)
    named = dict(fused.named_parameters())
    keys = [k for k in g if k.startswith(f"{vt}.grad.") and ".rays." not in k]
    assert len(keys) == 11
    for k in keys:
        name = k[len(vt) + 6:]
        want64 = g[k.replace(".grad.", ".grad64.")]
        bound, scale = grad_bound(g[k], want64, factor=4.0, floor=5e-3)
        err = float(np.abs(named[name].grad.detach().cpu().numpy().astype(np.float64) - want64).max())
        assert err <= bound, (vt, name, err, bound, scale)
    # the autograd path on the same batch
    auto = _model(scene_states["b"], cfg=cfg)
    out = auto(tb, is_training=True, background_rgb=bg, global_step=gs, _t_rand_primary=tp, _t_rand_shadow=ts)
    la = train_loss_dict(out, gt, 0.1)
    la["loss"].backward()
    np.testing.assert_allclose(float(loss8[0]), float(la["loss"]), rtol=2e-5)
    for (name, pa), (_, pf) in zip(auto.named_parameters(), fused.named_parameters()):
        scale = float(pa.grad.abs().max()) + 1e-30
        assert float((pa.grad - pf.grad).abs().max()) < 3e-4 * scale + 5e-7, (vt, name, float((pa.grad - pf.grad).abs().max()) / scale)
    # the hint gradients are really in: d loss / d variance differs from the hint-constant model's as the fixture says (shadow),
    # the first SDF layer's by ~10 % (specular)
    plain = _model(scene_states["b"])
    train_fused.train_step_backward(plain, tb, gt, bg, gs, t_rand_primary=tp, t_rand_shadow=ts)
    pn = dict(plain.named_parameters())
    if vt == "shg":
        a, b = named["deviation_network.variance"].grad, pn["deviation_network.variance"].grad
        assert abs(float(a - b)) > 0.5 * abs(float(a))
    else:
        a, b = named["sdf_network.lin0.weight_v"].grad, pn["sdf_network.lin0.weight_v"].grad
        assert float((a - b).abs().max()) > 1e-2 * float(b.abs().max())
    # captured
    eager = {k: v.grad.detach().clone() for k, v in fused.named_parameters()}
    cap = _model(scene_states["b"], cfg=cfg)
    step = GraphedTrainStep(cap, tb.origins.shape[0], bg, lr=0.0, warm_up_end=0, global_step=gs, jitter=(tp, ts), fused=True)
    try:
        assert step._use_fused
        got = step(tb, gt, global_step=gs)
        np.testing.assert_allclose(float(got["loss"]), float(loss8[0]), rtol=1e-6)
        for k, v in cap.named_parameters():
            scale = float(eager[k].abs().max()) + 1e-30
            assert float((v.grad - eager[k]).abs().max()) <= 1e-6 * scale, (vt, k)
    finally:
        step.release()
