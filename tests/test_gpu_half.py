"""16-bit hand-offs of the weight-gradient operands (include/nrhints_hip.h: nrh_sdf_train_forward_half / _backward_half,
NrhDwJob.half_ops; CHANGELOG.md section 7i) at the level of the C entry points: the fp16 arrays ARE the float32 arrays rounded to nearest, the
adjoint scale follows the seeds (exact invariance under a power-of-two change of their magnitude), and the weight gradients they
give agree with the float32 hand-offs to the operands' rounding.  The step-level parity - every gradient against the reference's
float64 step at 1 024 rays - is tests/test_gpu_train1024.py, which runs with the hand-offs on."""
import os

import numpy as np
import pytest
import torch

import nrhints_amd as na
from nrhints_amd import _lib, dw, ops

pytestmark = pytest.mark.gpu
# build-time experiment variant of the library (make variant DEFS=-DNRH_COUP16=1, csrc/nrh_api.hip coup16_mode): the default build has it off
COUP16 = "NRH_COUP16=1" in _lib.load().nrh_build_info().decode()
T = torch.from_numpy
NPTS = 32768          # the 8-wave kernels; 16 384 points = the 4-wave builds of the same source (csrc/nrh_small.hip) on 256 CUs


@pytest.fixture(scope="module", params=[NPTS, 16384])
def net(scene_states, request):
    NPTS = request.param
    model = na.NeuSHintRenderer(na.NeuSModelConfig(), precision="f16x3")
    model.load_state_dict({k: T(np.asarray(v)) for k, v in scene_states["b"].items()})
    pk = model.cuda().eval().packed_params(torch.device("cuda", torch.cuda.current_device()))
    g = torch.Generator().manual_seed(5)
    pts = ((torch.rand(NPTS, 3, generator=g) * 2 - 1) * 0.8).cuda()
    # adjoint-like seeds: heavy-tailed over the points, magnitudes of a 1 024-ray step
    amp = torch.exp(torch.randn(NPTS, 1, generator=g) * 2.0) * 1e-4
    sbar = (torch.randn(NPTS, generator=g) * amp[:, 0]).cuda()
    gbar = (torch.randn(NPTS, 3, generator=g) * amp * 0.1).cuda()
    fbar = (torch.randn(NPTS, 256, generator=g) * amp * 1e-3).cuda()
    return pk, pts, sbar, fbar, gbar


def _backward(pk, pts, sv, sbar, fbar, gbar, **kw):
    z3, t0 = torch.zeros(pts.shape[0], 3, device="cuda"), torch.zeros(pts.shape[0], 1, device="cuda")
    return ops.sdf_train_backward(pk["sdf_w"], pk["sdf_wt_feat"], pk["sdf_head"], pts, z3, t0, 1, sv, sbar, fbar, gbar, **kw)


def test_half_supported_predicate():
    lib = _lib.load()
    assert lib.nrh_train_half_supported(1, NPTS) == 1 and lib.nrh_train_half_supported(1, 131072) == 1
    assert lib.nrh_train_half_supported(0, NPTS) == 0          # exact-fp32 mode keeps float32 hand-offs
    assert lib.nrh_train_half_supported(1, 16384) == 1         # the 4-wave builds (csrc/nrh_small.hip): the same source
    assert lib.nrh_train_half_supported(1, 8192) == 0          # the channel-split kernels: not there
    assert lib.nrh_train_half_supported(1, NPTS + 16) == 0     # 32-point stages


def test_forward_half_arrays_are_the_rounded_float32_arrays(net):
    pk, pts = net[0], net[1]
    sdf, feat, grad, sv = ops.sdf_train_forward(pk["sdf_w"], pk["sdf_b"], pk["sdf_head"], pts)
    sdf_h, feat_h, grad_h, svh = ops.sdf_train_forward(pk["sdf_w"], pk["sdf_b"], pk["sdf_head"], pts, half_handoffs=True)
    assert torch.equal(sdf, sdf_h) and torch.equal(feat, feat_h) and torch.equal(grad, grad_h)
    assert torch.equal(sv["s1"], svh["s1"]) and torch.equal(sv["t"], svh["t"]) and torch.equal(sv["h"][7], svh["h"][7])
    rm = dw.from_tiled if dw.arrays_tiled() else (lambda x: x)
    for l in range(7):
        assert torch.equal(dw.from_half_tiled(svh["h16"][l]), rm(sv["h"][l]).half()), l
    for l in range(1, 8):
        assert torch.equal(dw.from_half_tiled(svh["t16"][l]), rm(sv["t"][l]).half()), l


def test_backward_half_arrays_scale_and_invariance(net):
    """abar16 / zbar16 = round(S x the adjoint) with S = 2^(4 - ceil(log2 max |seed|)); the same sweep told that S as a constant
    writes exactly those values into the float32 arrays; seeds 2^-9 times smaller give the same fp16 bits and S 2^9 times larger;
    zero seeds give S = 1 and zeros."""
    pk, pts, sbar, fbar, gbar = net
    NPTS = pts.shape[0]
    _, _, _, sv = ops.sdf_train_forward(pk["sdf_w"], pk["sdf_b"], pk["sdf_head"], pts, half_handoffs=True)
    dyn = torch.zeros(4, device="cuda")
    r = _backward(pk, pts, sv, sbar, fbar, gbar, half_handoffs=True, dyn=dyn)
    S, IS = float(dyn[0]), float(dyn[1])
    m = max(float(sbar.abs().max()), float(gbar.abs().max()), float(fbar[::8].abs().max()))
    assert 8.0 <= m * S < 16.0 and S * IS == 1.0 and np.log2(S) == round(np.log2(S))
    assert float(dyn[2]) == 0.0 and float(dyn[3]) == 0.0                  # the work words are reset for the next step
    ref = _backward(pk, pts, sv, sbar, fbar, gbar, adj_scale=S)
    rm = dw.from_tiled if dw.arrays_tiled() else (lambda x: x)
    for l in range(7):
        assert torch.equal(dw.from_half_tiled(r["abar16"][l]), (rm(ref["abar"][l]) * S).half()), ("abar", l)
    assert torch.equal(r["abar"][7], ref["abar"][7]) and torch.equal(r["gebar"], ref["gebar"])
    if not COUP16:
        for l in range(1, 8):
            assert torch.equal(dw.from_half_tiled(r["zbar16"][l]), (rm(ref["zbar"][l]) * S).half()), ("zbar", l)
        assert torch.equal(r["zbar"][0], ref["zbar"][0]) and torch.equal(r["pbar"], ref["pbar"]) and torch.equal(r["coup"], ref["coup"])
    else:
        # the experiment build -DNRH_COUP16=1: coup - the sweeps' private hand-off - is fp16 as well (S x the value, half-tiled, in the same
        # buffer); the value sweep sees it rounded to 11 bits, so zbar agrees with the float32 hand-off's to that rounding only
        coup16 = r["coup"].view(-1).view(torch.float16)[: 8 * NPTS * 256].view(8, NPTS, 256)
        for l in range(8):
            assert torch.equal(dw.from_half_tiled(coup16[l]), (rm(ref["coup"][l]) * S).half()), ("coup", l)
        assert float(coup16.abs().max()) < 2.0 ** 13
        for l in range(1, 8):
            want = rm(ref["zbar"][l]) * S
            err = (dw.from_half_tiled(r["zbar16"][l]).float() - want).abs()
            assert float(err.max()) < 1e-3 * float(want.abs().max()), ("zbar", l, float(err.max()), float(want.abs().max()))
        for key, x, y in (("zbar0", r["zbar"][0], ref["zbar"][0]), ("pbar", r["pbar"], ref["pbar"])):
            assert float((x - y).abs().max()) < 1e-3 * float(y.abs().max()), key
    # the stored maxima sit well inside fp16's range
    top = max(float(r["abar16"][:7].abs().max()), float(r["zbar16"][1:].abs().max()))
    assert 2.0 ** -8 < top < 2.0 ** 10, top
    k = 2.0 ** -9
    r2 = _backward(pk, pts, sv, sbar * k, fbar * k, gbar * k, half_handoffs=True, dyn=dyn)
    assert float(dyn[0]) == S / k
    assert torch.equal(r2["abar16"][:7], r["abar16"][:7]) and torch.equal(r2["zbar16"][1:], r["zbar16"][1:])      # (the written layers)
    for key, x2, x1 in (("pbar", r2["pbar"], r["pbar"]), ("zbar0", r2["zbar"][0], r["zbar"][0]), ("gebar", r2["gebar"], r["gebar"])):
        bad = (x2 != x1 * k) & ((x1 * k).abs() > 1e-36)          # (float32 subnormals round differently under the second scaling)
        assert not bool(bad.any()), (key, int(bad.sum()), x2[bad][:4].tolist(), (x1 * k)[bad][:4].tolist())
    z = _backward(pk, pts, sv, sbar * 0, fbar * 0, gbar * 0, half_handoffs=True, dyn=dyn)
    assert float(dyn[0]) == 1.0 and float(z["abar16"][:7].abs().max()) == 0.0 and float(z["zbar16"][1:].abs().max()) == 0.0


def test_weight_gradients_from_half_handoffs(net):
    """nrh_dw_gemm on the SDF net's job table with the 16-bit hand-offs against the same table on the float32 arrays (bf16 x 3
    products): the difference is the operands' rounding to 11 bits - random, so ~2^-12 of sqrt(sum (a b)^2) per entry; asserted:
    1e-3 of the tensor's scale (seeds this heavy-tailed leave few effective terms per entry: 2.8e-4 measured at 32 768 points, 5.2e-4
    at 16 384; the 1 024-ray step shows 1.5e-4)."""
    pk, pts, sbar, fbar, gbar = net
    NPTS = pts.shape[0]
    _, _, _, sv = ops.sdf_train_forward(pk["sdf_w"], pk["sdf_b"], pk["sdf_head"], pts, half_handoffs=True)
    _, _, _, sv32 = ops.sdf_train_forward(pk["sdf_w"], pk["sdf_b"], pk["sdf_head"], pts)
    dyn = torch.zeros(4, device="cuda")
    r = _backward(pk, pts, sv, sbar, fbar, gbar, half_handoffs=True, dyn=dyn)
    r32 = _backward(pk, pts, sv32, sbar, fbar, gbar, adj_scale=float(dyn[0]))
    shapes = [(256, 39)] + [(217 if l == 3 else 256, 256) for l in range(1, 8)]

    def outs():
        new = lambda *s: torch.full(s, float("nan"), device="cuda")
        o = {f"dW{l}": new(*shapes[l]) for l in range(8)}
        o.update({f"db{l}": new(shapes[l][0]) for l in range(8)})
        o.update(ws=new(1, 256), bs=new(1), Wf=new(256, 256), bf=new(256))
        return o

    emb = torch.randn(NPTS, 64, device="cuda")
    a, b = outs(), outs()
    half = dict(h16=sv["h16"], t16=sv["t16"], zbar16=r["zbar16"], abar16=r["abar16"], dyn=dyn)
    dw.run(dw.sdf_jobs(shapes, sv["h"], sv["t"], r["zbar"], r["abar"], r["gebar"], emb, sbar, fbar, a, half=half), NPTS)
    dw.run(dw.sdf_jobs(shapes, sv32["h"], sv32["t"], r32["zbar"], r32["abar"], r32["gebar"], emb, sbar, fbar, b), NPTS)
    torch.cuda.synchronize()
    rm = dw.from_tiled if dw.arrays_tiled() else (lambda x: x)
    for l in range(8):
        assert bool(torch.isfinite(a[f"dW{l}"]).all()) and bool(torch.isfinite(a[f"db{l}"]).all()), l
        scale = float(b[f"dW{l}"].abs().max())
        err = (a[f"dW{l}"] - b[f"dW{l}"]).abs()
        if l == 0:
            # float32 hand-offs either way (another split of the points over the work items: fp32 summation order only)
            assert float(err.max()) < (1e-4 if COUP16 else 1e-5) * scale
            assert float((a["db0"] - b["db0"]).abs().max()) < (1e-4 if COUP16 else 1e-5) * float(b["db0"].abs().max())
            continue
        # (relative to the TENSOR's scale: entries of channels whose adjoints sit below fp16's absolute floor - 2^-24 / S - come
        # out as zero; they are < 1e-6 of the scale)
        assert float(err.max()) < 1e-3 * scale, (l, float(err.max()), scale)
        dberr = float((a[f"db{l}"] - b[f"db{l}"]).abs().max())
        assert dberr < 1e-3 * float(b[f"db{l}"].abs().max()) + 1e-30, (l, dberr)
    for k in ("ws", "bs", "Wf", "bf"):
        assert float((a[k] - b[k]).abs().max()) <= 1e-5 * float(b[k].abs().max()), k       # (summation order: another split of the points)
