"""Differentiable ``render_core``: what the loss differentiates, as autograd Functions over HIP kernels.

The graph-less stages (both hierarchical samplers, the 128-sample shadow march, depth / hit point, specular cue:
models/neus_hint_model.py:697, :531, :379, :589) run in ``nrh_render_forward_train``; this module wires the differentiable
part - SDF + feature + d sdf/dx at the 128 section mid-points, alpha compositing and the reflectance net (:504-510,
:521-525, :583-587, :626-637) - out of ``SdfValueFeatGradHip`` (sdf_function.py), ``AlphaWeightsNormalsHip`` and
``ColorNetHip`` (forward and adjoint kernels, dW as rocBLAS split-K GEMMs).  What is left to torch ops: the per-ray
encodings (tiny; they carry the ray gradients for pose refinement), ``rgb = sum w c + bg (1 - sum w)`` and the loss.
It is only entered when a gradient is requested, and only on the GPU.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

from .sdf_function import sdf_value_feat_grad


_FREQS = {}


def _freqs(n_freq: int, like: torch.Tensor) -> torch.Tensor:
    """2^k, k < n_freq, cached per (device, dtype): a training step encodes a dozen tensors."""
    key = (n_freq, str(like.device), like.dtype)
    if key not in _FREQS:
        _FREQS[key] = 2.0 ** torch.arange(n_freq, dtype=like.dtype, device=like.device)
    return _FREQS[key]


def _enc(x: torch.Tensor, n_freq: int) -> torch.Tensor:
    """NeRF encoding with include_input (fields/encodings.py:168-174)."""
    s = (x[..., None] * _freqs(n_freq, x)).reshape(*x.shape[:-1], -1)
    return torch.cat([x, torch.sin(torch.cat([s, s + math.pi / 2.0], dim=-1))], dim=-1)


class AlphaWeightsNormalsHip(torch.autograd.Function):
    """(sdf [P,1], grad [P,3], dirs [N,3], dists [N,128], variance) -> (weights [N,128], n_hat [P,3]) with the forward
    and the adjoint in one HIP kernel each (csrc/nrh_rays_train.hip) - get_alpha, the exclusive transmittance product
    and F.normalize of models/neus_hint_model.py:339-356, :521-525, :584 and what autograd derives from them."""

    @staticmethod
    def forward(ctx, sdf, grad, dirs, dists, variance, inv_s: float, cos_anneal: float, dyn=None, n_real: int = 128):
        from . import _lib
        lib = _lib.load()
        n = dirs.shape[0]
        f32c = lambda t: t.detach().to(torch.float32).contiguous()
        sdf_c, grad_c, dirs_c, dists_c = f32c(sdf), f32c(grad), f32c(dirs), f32c(dists)
        weights = torch.empty(n, 128, dtype=torch.float32, device=dirs.device)
        nhat = torch.empty(n * 128, 3, dtype=torch.float32, device=dirs.device)
        P = _lib.ptr
        _lib.check(lib.nrh_alpha_train_forward_n(P(sdf_c), P(grad_c), P(dirs_c), P(dists_c), float(inv_s), float(cos_anneal), P(dyn), n,
                                                 int(n_real), P(weights), P(nhat), _lib.stream_handle()), "nrh_alpha_train_forward")
        ctx.save_for_backward(sdf_c, grad_c, dirs_c, dists_c)
        ctx.consts = (float(inv_s), float(cos_anneal))
        ctx.n_real = int(n_real)
        ctx.dyn = dyn
        return weights, nhat

    @staticmethod
    def backward(ctx, wbar, nbar):
        from . import _lib
        lib = _lib.load()
        sdf_c, grad_c, dirs_c, dists_c = ctx.saved_tensors
        inv_s, cos_anneal = ctx.consts
        n = dirs_c.shape[0]
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dirs_c.device)
        wbar = torch.zeros(n, 128, dtype=torch.float32, device=dirs_c.device) if wbar is None else wbar.to(torch.float32).contiguous()
        nbar = None if nbar is None else nbar.to(torch.float32).contiguous()
        sdf_bar, grad_bar, rd_bar, invs_bar = new(n * 128, 1), new(n * 128, 3), new(n, 3), new(n)
        P = _lib.ptr
        dyn = ctx.dyn
        _lib.check(lib.nrh_alpha_train_backward_n(P(sdf_c), P(grad_c), P(dirs_c), P(dists_c), inv_s, cos_anneal, P(dyn), n, ctx.n_real,
                                                  P(wbar), P(nbar), P(sdf_bar), P(grad_bar), P(rd_bar), P(invs_bar), _lib.stream_handle()),
                   "nrh_alpha_train_backward")
        # inv_s = clip(exp(10 variance), 1e-6, 1e6): d inv_s / d variance = 10 inv_s inside the clip range
        if dyn is not None:       # device-side inv_s (hipGraph mode): same chain rule without a host value
            s_dev = dyn[0]
            var_bar = invs_bar.sum() * torch.where((s_dev > 1e-6) & (s_dev < 1e6), 10.0 * s_dev, torch.zeros_like(s_dev))
        else:
            var_bar = invs_bar.sum() * (10.0 * inv_s if 1e-6 < inv_s < 1e6 else 0.0)
        return sdf_bar, grad_bar, rd_bar, None, var_bar, None, None, None, None


class ColorNetHip(torch.autograd.Function):
    """Reflectance net for 128 samples per ray with the forward and the adjoint sweep in the HIP register-chain kernels
    (csrc/nrh_color.hip, TRAIN instantiation + color_adjoint_kernel) and the weight gradients as split-K GEMMs.
    forward(feat [P,256], pts [P,3], normal [P,3], ray_enc [N,99|54], packed, w0..w4, b0..b4) -> colour [P,3]
    (ray_enc = cat[enc4(view), enc4(pl) (, enc4(vis), enc4(cue))] per ray; its adjoint is returned per ray, so the
    caller's autograd carries it on to the rays)."""

    @staticmethod
    def forward(ctx, feat, pts, normal, ray_enc, packed, *params):
        from . import _lib, packing
        lib = _lib.load()
        hints = bool(packed["hints"])
        dev = feat.device
        Pn = feat.shape[0]
        n = Pn // 128
        rows = ray_enc.shape[0]       # one per ray, or one per group of 128 / clip samples (partial visibility hint)
        f32c = lambda t: t.detach().to(torch.float32).contiguous()
        feat_c, pts_c, nrm_c = f32c(feat), f32c(pts), f32c(normal)
        raymisc = torch.zeros(rows, packing.RAYMISC_STRIDE, dtype=torch.float32, device=dev)
        raymisc[:, :ray_enc.shape[1]] = ray_enc.detach()
        mw = 128 if hints else 64
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        color, save_h, save_misc = new(Pn, 3), new(4, Pn, 256), new(Pn, mw)
        P = _lib.ptr
        cw = packed["col_w"]
        _lib.check(lib.nrh_color_train_forward_grouped(packed["precision"], int(hints), P(cw, cw.dtype), P(packed["col_b"]), P(feat_c),
                                                       P(pts_c), P(nrm_c), P(raymisc), Pn // rows, n, P(color), P(save_h), P(save_misc),
                                                       _lib.stream_handle()), "nrh_color_train_forward")
        ctx.save_for_backward(feat_c, color, save_h, save_misc)
        ctx.packed, ctx.n, ctx.enc_width, ctx.rows = packed, n, ray_enc.shape[1], rows
        ctx.shapes = [tuple(t.shape) for t in params]
        return color

    @staticmethod
    def backward(ctx, cbar):
        from . import _lib, packing
        lib = _lib.load()
        feat_c, color, save_h, save_misc = ctx.saved_tensors
        packed, n = ctx.packed, ctx.n
        hints = bool(packed["hints"])
        Pn = n * 128
        dev = feat_c.device
        mw = save_misc.shape[1]
        zbar4 = (cbar.to(torch.float32) * color * (1.0 - color)).contiguous()        # through the sigmoid
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        zbar, fbar, mbar = new(4, Pn, 256), new(Pn, 256), new(Pn, mw)
        P = _lib.ptr
        cwt = packed["col_wt"]
        _lib.check(lib.nrh_color_train_backward(packed["precision"], int(hints), P(cwt, cwt.dtype), P(zbar4), P(save_h), n, P(zbar),
                                                P(fbar), P(mbar), _lib.adjoint_scale_from_seeds((zbar4,), n), _lib.stream_handle()),
                   "nrh_color_train_backward")
        nm = 105 if hints else 60
        grads_in = (fbar, mbar[:, 0:3], mbar[:, 3:6], mbar[:, 6:nm].reshape(ctx.rows, Pn // ctx.rows, nm - 6).sum(1), None)
        if not any(ctx.needs_input_grad[5:]):
            return grads_in + (None,) * 10
        # weight gradients: one split-K bf16x3 MFMA launch (csrc/nrh_dw.hip), bias gradients = its column sums
        from . import dw
        out = {f"w{l}": new(*ctx.shapes[l]) for l in range(5)}
        out.update({f"b{l}": new(*ctx.shapes[5 + l]) for l in range(5)})
        dw.run(dw.color_jobs(hints, zbar, zbar4, save_h, feat_c, save_misc, out), Pn)
        return grads_in + tuple(out[f"w{l}"] for l in range(5)) + tuple(out[f"b{l}"] for l in range(5))


class ShadowVisibilityHip(torch.autograd.Function):
    """(sdf [P,1], grad [P,3], shadow_dirs [N,3], dists [N,128], variance) -> visibility [N,1] = transmittance in front of the
    shadow ray's last sample (get_alpha + the cumprod of models/neus_hint_model.py:339-356, :428-432), forward and adjoint in the
    alpha-stage kernel pair (csrc/nrh_rays_train.hip, nrh_shadow_alpha_*).  For renderer.shadow_hint_gradient."""

    @staticmethod
    def forward(ctx, sdf, grad, dirs, dists, variance, inv_s: float, cos_anneal: float, dyn=None, n_real: int = 128):
        from . import _lib
        lib = _lib.load()
        n = dirs.shape[0]
        f32c = lambda t: t.detach().to(torch.float32).contiguous()
        sdf_c, grad_c, dirs_c, dists_c = f32c(sdf), f32c(grad), f32c(dirs), f32c(dists)
        vis = torch.empty(n, 1, dtype=torch.float32, device=dirs.device)
        P = _lib.ptr
        _lib.check(lib.nrh_shadow_alpha_forward(P(sdf_c), P(grad_c), P(dirs_c), P(dists_c), float(inv_s), float(cos_anneal), P(dyn), n,
                                                int(n_real), P(vis), _lib.stream_handle()), "nrh_shadow_alpha_forward")
        ctx.save_for_backward(sdf_c, grad_c, dirs_c, dists_c)
        ctx.consts, ctx.dyn, ctx.n_real = (float(inv_s), float(cos_anneal)), dyn, int(n_real)
        return vis

    @staticmethod
    def backward(ctx, vbar):
        from . import _lib
        lib = _lib.load()
        sdf_c, grad_c, dirs_c, dists_c = ctx.saved_tensors
        inv_s, cos_anneal = ctx.consts
        n = dirs_c.shape[0]
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dirs_c.device)
        vbar = vbar.to(torch.float32).contiguous()
        sdf_bar, grad_bar, rd_bar, invs_bar = new(n * 128, 1), new(n * 128, 3), new(n, 3), new(n)
        P = _lib.ptr
        dyn = ctx.dyn
        _lib.check(lib.nrh_shadow_alpha_backward(P(sdf_c), P(grad_c), P(dirs_c), P(dists_c), inv_s, cos_anneal, P(dyn), n, ctx.n_real,
                                                 P(vbar), P(sdf_bar), P(grad_bar), P(rd_bar), P(invs_bar), _lib.stream_handle()),
                   "nrh_shadow_alpha_backward")
        if dyn is not None:
            s_dev = dyn[0]
            var_bar = invs_bar.sum() * torch.where((s_dev > 1e-6) & (s_dev < 1e6), 10.0 * s_dev, torch.zeros_like(s_dev))
        else:
            var_bar = invs_bar.sum() * (10.0 * inv_s if 1e-6 < inv_s < 1e6 else 0.0)
        return sdf_bar, grad_bar, rd_bar, None, var_bar, None, None, None, None


def _specular_cue(hit_normal, pl, hit, dirs, roughness) -> torch.Tensor:
    """Cook-Torrance specular cue per roughness (models/neus_hint_model.py:589-615), differentiable in the hit normal, the
    light position and the view direction; [N,3] inputs -> [N, len(roughness)]."""
    lit = F.normalize(pl - hit, dim=-1, p=2)
    view = F.normalize(-dirs, dim=-1, p=2)
    half = F.normalize(lit + view, dim=-1, p=2)
    n_l = (hit_normal * lit).sum(-1).clip(0.0, 1.0)
    n_v = (hit_normal * view).sum(-1).clip(0.0, 1.0)
    n_h = (hit_normal * half).sum(-1).clip(0.0, 1.0)
    h_v = (half * view).sum(-1).clip(0.0, 1.0)
    n_h2 = torch.pow(n_h, 2)
    out = []
    for r in roughness:
        k = (r + 1.0) * (r + 1.0) / 8.0
        g = (n_v / (n_v * (1.0 - k) + k)) * (n_l / (n_l * (1.0 - k) + k))
        a2 = r * r
        ndf = a2 / (torch.pi * torch.pow(n_h2 * (a2 - 1.0) + 1.0, 2))
        f = 0.04 + 0.96 * torch.pow(1.0 - h_v, 5)
        out.append(ndf * g * f / (4.0 * n_v + 1e-3))
    return torch.stack(out, dim=-1)


def _visibility(d, packed, variance, pl, hit, mid_z, dists, cos_anneal: float, dyn=None, n_real: int = 128) -> torch.Tensor:
    """The differentiable tail of get_visibility (models/neus_hint_model.py:411-432) for renderer.shadow_hint_gradient: alpha at
    the 128 section mid-points of the shadow ray light -> hit point (sections from the graph-less HIP sampler; the reference
    detaches its importance samples too, :313), transmittance in front of the last one.  The SDF network and d sdf/dx at the
    shadow points run through the same HIP forward / backward sweeps as the primary samples (sdf_value_feat_grad), the alpha
    stage through its kernel pair (ShadowVisibilityHip)."""
    n, T = mid_z.shape
    sd = hit - pl
    srd = sd / torch.linalg.norm(sd, ord=2, dim=-1, keepdim=True)
    pts = (pl[:, None, :] + srd[:, None, :] * mid_z[..., None]).reshape(-1, 3)
    sdf, _, grad = sdf_value_feat_grad(d, pts, packed=packed)
    # n_real: the shadow ray's samples that exist (n_shadow_samples + its importance samples; 128 with the defaults): the visibility
    # is the transmittance in front of the last of THOSE, padded slots carry nothing
    return ShadowVisibilityHip.apply(sdf, grad, srd, dists, variance, packed["inv_s"], cos_anneal, dyn, n_real)


def render_core(d: Dict[str, torch.Tensor], variance: torch.Tensor, o, dirs, pl, mid_z, dists, vis, cue, cos_anneal: float,
                background_rgb, analytic_normal: bool = False, packed=None, pre=None, dyn=None, hint_grad=None,
                n_real: int = 128) -> Dict[str, torch.Tensor]:
    """``d``: weight-norm-folded dense parameters WITH autograd history (packing.dense_params* on the live nn.Parameters);
    mid_z / dists [N,128], vis [N,1], cue [N,4]: graph-less results of the HIP forward; ``packed``: the kernel buffers of the
    same parameters.  Every network evaluation and its adjoint is a HIP kernel (there is no torch formulation in here; the
    ones the kernels are tested against are tests/torch_backends.py).  ``hint_grad`` (renderer.shadow_hint_gradient /
    specular_hint_gradient, off by default): dict(hit, specular, shadow, roughness) - the hints are then re-derived here as
    differentiable functions of the network (and of the rays) instead of entering as constants."""
    if packed is None or packed.get("col_wt") is None:
        raise ValueError("render_core needs the packed parameters incl. the transposed reflectance weights")
    n, T = mid_z.shape
    pts = (o[:, None, :] + dirs[:, None, :] * mid_z[..., None]).reshape(-1, 3)
    sdf, feat, grad = sdf_value_feat_grad(d, pts, packed=packed, pre=pre)
    inv_s = torch.exp(variance * 10.0).clip(1e-6, 1e6)
    # alpha, transmittance product, weights and unit normals: one HIP kernel forward, one for the adjoint
    # n_real = 64: renderer.n_importance_samples = 0 - slots 64..127 of every ray are padding with weight exactly 0
    weights, n_hat = AlphaWeightsNormalsHip.apply(sdf, grad, dirs, dists, variance, packed["inv_s"], cos_anneal, dyn, n_real)
    if hint_grad is not None:
        if hint_grad.get("shadow") is not None:
            vis = _visibility(d, packed, variance, pl, hint_grad["hit"], hint_grad["shadow"]["mid_z"], hint_grad["shadow"]["dists"],
                              cos_anneal, dyn, int(hint_grad["shadow"].get("n_real", 128)))
        if hint_grad.get("specular"):
            hit_n = F.normalize((n_hat.reshape(n, T, 3) * weights[..., None]).sum(1), dim=-1, p=2)      # :586-587
            cue = _specular_cue(hit_n, pl, hint_grad["hit"], dirs, hint_grad["roughness"])
    # per-ray part of the reflectance input: encodings of view direction, light position, visibility hint, specular cue
    per_ray = [_enc(dirs, 4), _enc(pl, 4)]
    if vis is not None:  # vis / cue are None for the pl-naive model (no hints)
        per_ray += [_enc(vis, 4), _enc(cue, 4)]
        clip = vis.shape[0] // n
        if clip > 1:     # partial visibility hint: vis [N * clip, 1], one row per group of 128 / clip samples (:553-570)
            per_ray = [t if t.shape[0] == n * clip else t.repeat_interleave(clip, dim=0) for t in per_ray]
    normal = grad if analytic_normal else n_hat
    # reflectance net: forward and adjoint sweep in the HIP register-chain kernels, dW as split-K GEMMs
    col = ColorNetHip.apply(feat, pts, normal, torch.cat(per_ray, dim=-1), packed,
                            *[d[f"col_w{l}"] for l in range(5)], *[d[f"col_b{l}"] for l in range(5)]).reshape(n, T, 3)
    rgb = (col * weights[..., None]).sum(1)
    if background_rgb is not None:
        rgb = rgb + background_rgb * (1.0 - weights.sum(-1, keepdim=True))
    return dict(rgb=rgb, weights=weights, analytic_normals=grad.reshape(n, T, 3),
                normalized_analytic_normals=n_hat.reshape(n, T, 3), s_val=(1.0 / inv_s).expand(n, T))
