"""``SdfValueFeatGradHip`` - a torch.autograd.Function for (sdf, feature, d sdf/dx) of the SDF network with a HAND-DERIVED
backward in HIP kernels, so that the training step needs no second-order autograd graph.

Why: the loss reaches the network through three outputs - the value (alpha), the feature (colour net) and the spatial
gradient g = d sdf/dx (alpha's cosine, the unit normal fed to the colour net, the eikonal term).  The reference obtains
g with ``autograd.grad(create_graph=True)`` and lets autograd differentiate that graph again
(fields/sdf_field.py:136-148, pipelines/base_pipeline.py:59-62).  Here the second-order part is written out:

    <g_bar, g> is a JVP of the network in direction g_bar, so its parameter gradient is obtained by
      * one FORWARD-direction sweep of "tangent adjoints"   tbar_l = W_l abar_l,   abar_{l+1} = s'_l * tbar_l
      * one REVERSE sweep of value adjoints with a coupling term   zbar_l = s'_l * hbar_l + s''_l * a_{l+1} * tbar_l
      * two plain GEMMs per layer for the weights:   dW_l = zbar_l^T x_l + t_l^T abar_l        (t_l = s'_l * a_{l+1})

with s' = sigmoid(100 z), s'' = 100 s'(1 - s'), a_l the reverse-chain values of the forward pass.  Forward and both sweeps
are the HIP register-chain kernels (csrc/nrh_sdf.hip MODE 3, csrc/nrh_sdf_train.hip); the weight gradients are one split-K
bf16x3 MFMA launch over the saved arrays - in the sweeps' TILED layout (include/nrhints_hip.h, "LAYOUT OF THE [8][npts][256] ARRAYS"), which
nrh_dw_gemm reads natively - (csrc/nrh_dw.hip through nrhints_amd/dw.py; no library GEMM).  The same maths in plain torch ops - the reference the kernels are tested against -
lives with the tests (tests/torch_backends.py), not in the product.
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch
import torch.nn.functional as F

N_LAYERS = 8
SKIP = 4
EMB = 39


_ENC_CONST = {}


def _enc_parts(x3: torch.Tensor):
    """Encoding entries e(x3) [P,39], first derivatives dc [P,39] and second derivatives d2c [P,39] w.r.t. the
    coordinate each entry depends on; ``dim`` [39] says which coordinate that is (fields/encodings.py:168-174)."""
    key = (str(x3.device), x3.dtype)
    if key not in _ENC_CONST:
        freqs = 2.0 ** torch.arange(6, dtype=x3.dtype, device=x3.device)
        dim = torch.cat([torch.arange(3), torch.arange(3).repeat_interleave(6), torch.arange(3).repeat_interleave(6)]).to(x3.device)
        _ENC_CONST[key] = (freqs, freqs.repeat(3), dim)
    freqs, fr, dim = _ENC_CONST[key]
    s = (x3[..., None] * freqs).reshape(x3.shape[0], -1)          # [P,18] d-major, k-minor
    sp = s + math.pi / 2.0
    e = torch.cat([x3, torch.sin(s), torch.sin(sp)], dim=1)
    dc = torch.cat([torch.ones_like(x3), torch.cos(s) * fr, torch.cos(sp) * fr], dim=1)
    d2c = torch.cat([torch.zeros_like(x3), -torch.sin(s) * fr * fr, -torch.sin(sp) * fr * fr], dim=1)
    return e, dc, d2c, dim


def _scatter_dims(v: torch.Tensor, dim: torch.Tensor = None) -> torch.Tensor:
    """[P,39] -> [P,3]: sum the entries that depend on each coordinate.  The encoding is laid out
    [x (3) | sin, coordinate-major x 6 frequencies (18) | cos, same (18)], so this is three reshaped sums."""
    P = v.shape[0]
    return v[:, 0:3] + v[:, 3:21].reshape(P, 3, 6).sum(-1) + v[:, 21:39].reshape(P, 3, 6).sum(-1)


class SdfValueFeatGradHip(torch.autograd.Function):
    """Same contract as ``SdfValueFeatGrad`` with the forward and both backward sweeps in the HIP register-chain
    kernels (csrc/nrh_sdf.hip MODE 3, csrc/nrh_sdf_train.hip); the weight gradients are jobs of nrh_dw_gemm over the saved
    (tiled) arrays.  ``packed``: dict(sdf_w, sdf_b, sdf_head, sdf_wt_feat) packed from the SAME dense weights."""

    @staticmethod
    def forward(ctx, pts, packed, pre, *params):
        from . import ops
        if not pts.is_cuda:
            raise RuntimeError("sdf_backward='hip' needs the points on the GPU (no CPU fallback)")
        n = pts.shape[0]
        ctx.shapes = [tuple(t.shape) for t in params]
        if pre is not None:
            # the renderer's fused training call already evaluated the network at exactly these points
            # (pts = ro + rd * t, same fp32 expression): adopt its outputs and saved arrays
            ctx.packed, ctx.saves, ctx.n = packed, pre["saves"], n
            ctx.rays = (pre["ro"], pre["rd"], pre["t"], pre["n_per_ray"])
            ctx.p = pts.detach()
            return pre["sdf"], pre["feat"], pre["grad"]
        pad = (-n) % 32          # the sweep kernels take multiples of 16 points, the weight-gradient kernel multiples of 32
        p = pts.detach().to(torch.float32)
        if pad:
            p = torch.cat([p, p[-1:].expand(pad, 3)], dim=0)
        p = p.contiguous()
        sdf, feat, g, saves = ops.sdf_train_forward(packed["sdf_w"], packed["sdf_b"], packed["sdf_head"], p)
        ctx.packed, ctx.saves, ctx.p, ctx.n = packed, saves, p, n
        ctx.rays = (p, saves["zeros3"], saves["zeros1"], 1)
        return sdf[:n], feat[:n], g[:n]

    @staticmethod
    def backward(ctx, sbar, fbar, gbar):
        from . import ops
        packed, saves, p, n = ctx.packed, ctx.saves, ctx.p, ctx.n
        m = p.shape[0]

        def full(x, width):
            if x is not None and m == n:
                return x.reshape(n, width).to(torch.float32).contiguous()
            out = torch.zeros(m, width, dtype=torch.float32, device=p.device)
            if x is not None:
                out[:n] = x.reshape(n, width)
            return out

        sb, fb, gb = full(sbar, 1), full(fbar, 256), full(gbar, 3)
        ro, rd, tt, npr = ctx.rays
        # the adjoint scale follows the incoming adjoints (whatever the caller's loss normalisation; _lib.adjoint_scale_from_seeds)
        from . import _lib as _l
        r = ops.sdf_train_backward(packed["sdf_w"], packed["sdf_wt_feat"], packed["sdf_head"], ro, rd, tt, npr, saves,
                                   sb.reshape(-1), fb, gb, adj_scale=_l.adjoint_scale_from_seeds((sb, fb, gb), m // 128))
        zbar, abar, h, t = r["zbar"], r["abar"], saves["h"], saves["t"]
        p_bar = None
        if ctx.needs_input_grad[0]:       # only pose / light refinement asks for the points' adjoint
            e, dc, d2c, dim = _enc_parts(p * 3.0)
            ge = saves["ge"][:, :EMB] + saves["ge"][:, 73:73 + EMB]
            p_bar = (r["pbar"] + 9.0 * gb * _scatter_dims(ge * d2c))[:n]   # gbar[dim(e)] is constant within a coordinate's entries
        if not any(ctx.needs_input_grad[3:]):       # frozen network (e.g. register_view): only the points' adjoint
            ctx.saves = None
            return (p_bar, None, None) + (None,) * 20
        # weight gradients: [256 x P] @ [P x 256] products with a tiny output and a huge reduction - one split-K bf16x3 MFMA
        # launch for all layers and heads (csrc/nrh_dw.hip, nrhints_amd/dw.py), the bias gradients are its column sums
        from . import _lib, dw
        dev = p.device
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        out = {}
        for l in range(N_LAYERS):
            out[f"dW{l}"], out[f"db{l}"] = new(*ctx.shapes[l]), new(ctx.shapes[l][0])
        out["ws"], out["bs"], out["Wf"], out["bf"] = new(1, 256), new(1), new(256, 256), new(256)
        emb = new(m, 64)
        lib = _lib.load()
        with torch.cuda.device(dev):
            _lib.check(lib.nrh_embedding_rows(_lib.ptr(ro), _lib.ptr(rd), _lib.ptr(tt), npr, npr, ro.shape[0], _lib.ptr(emb), _lib.stream_handle()),
                       "nrh_embedding_rows")
            dw.run(dw.sdf_jobs(ctx.shapes[:N_LAYERS], h, t, zbar, abar, r["gebar"], emb, sb.reshape(-1), fb, out), m)
        ctx.saves = None
        return (p_bar, None, None, *[out[f"dW{l}"] for l in range(N_LAYERS)], *[out[f"db{l}"] for l in range(N_LAYERS)],
                out["ws"], out["bs"], out["Wf"], out["bf"])


def sdf_value_feat_grad(dense: Dict[str, torch.Tensor], pts: torch.Tensor, packed, pre=None):
    """(sdf, feat, d sdf/dx) at ``pts`` through the HIP kernels, differentiable w.r.t. the points and the dense
    (weight-norm-folded) weights of ``packing.dense_params``; ``packed``: the kernel buffers packed from those same weights."""
    if packed is None:
        raise ValueError("sdf_value_feat_grad needs the packed parameters (the HIP kernels are the only implementation)")
    args = [dense[f"sdf_w{l}"] for l in range(8)] + [dense[f"sdf_b{l}"] for l in range(8)] + \
           [dense["sdf_head_w"], dense["sdf_head_b"], dense["feat_w"], dense["feat_b"]]
    return SdfValueFeatGradHip.apply(pts, packed, pre, *args)
