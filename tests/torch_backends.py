"""Torch-only formulations of the differentiable render core - TEST INFRASTRUCTURE, not product code.

The product (nrhints_amd/autograd_core.py, sdf_function.py) evaluates both networks and their adjoints in HIP kernels only.
The formulations the kernels are checked against live here:

  * ``SdfValueFeatGrad``: the hand-derived tangent / value sweeps of the SDF net in torch ops (the maths the HIP sweeps
    implement, see nrhints_amd/sdf_function.py's header) - "manual";
  * the reference's own formulation, a second-order autograd graph through ``autograd.grad(create_graph=True)`` - "autograd";
  * the reflectance net by column blocks of its first layer and the alpha / weights / normals expressions in torch ops.

``render_core_torch`` has the signature of ``autograd_core.render_core`` plus ``sdf_impl``; tests swap it in with
``use_torch_backend(model, impl)``.
"""
from __future__ import annotations

import contextlib
import math
from typing import Dict

import torch
import torch.nn.functional as F

from nrhints_amd import autograd_core
from nrhints_amd.autograd_core import _enc
from nrhints_amd.sdf_function import EMB, N_LAYERS, SKIP, _enc_parts, _scatter_dims  # noqa: F401


# ---- helpers of the torch formulations (test infrastructure: library GEMMs and torch reductions live here, not in the product) ----
class _LinearBigK(torch.autograd.Function):
    """F.linear whose weight gradient ([out x P] @ [P x in], P ~ 1e5 rows, tiny output) is computed as a batched GEMM
    over S slabs of rows + a sum: the plain GEMM autograd would call launches ~32 workgroups on a 256-CU GPU."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.addmm(b, x, w.t())

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        m = x.shape[0]
        S = math.gcd(m, 64)
        gw = torch.bmm(gy.reshape(S, m // S, -1).transpose(1, 2), x.reshape(S, m // S, -1)).sum(0)
        return gy @ w, gw, gy.sum(0)


def _linear(x, w, b):
    return _LinearBigK.apply(x, w, b) if (x.is_cuda and x.shape[0] >= 4096) else F.linear(x, w, b)



_COL_INDEX = {}


def _col_index(device, hints: bool):
    """Column indices of the per-ray and of the (pts, normal) blocks in the reference's reflectance input, per device."""
    key = (str(device), hints)
    if key not in _COL_INDEX:
        cols = [torch.arange(3, 30), torch.arange(33, 60)] + ([torch.arange(316, 325), torch.arange(325, 361)] if hints else [])
        _COL_INDEX[key] = (torch.cat(cols).to(device), torch.tensor([0, 1, 2, 30, 31, 32], device=device))
    return _COL_INDEX[key]



def _colsum(x3: torch.Tensor) -> torch.Tensor:
    """[L,P,C] -> [L,C] column sums in two stages (64 slabs of rows first): the one-stage reduction of a [8,131072,256]
    array ran at 1.6 TB/s."""
    L, P, C = x3.shape
    S = math.gcd(P, 64)
    return x3.reshape(L, S, P // S, C).sum(2).sum(1)




class SdfValueFeatGrad(torch.autograd.Function):
    """forward(pts [P,3], W0..W7, b0..b7, ws [1,256], bs [1], Wf [256,256], bf [256]) -> sdf [P,1], feat [P,256], g [P,3].
    Weights are the dense (weight-norm-folded) matrices of ``packing.dense_params``; W4 is the UNSCALED layer-4 matrix."""

    @staticmethod
    def forward(ctx, pts, *params):
        W: List[torch.Tensor] = list(params[0:8])
        b: List[torch.Tensor] = list(params[8:16])
        ws, bs, Wf, bf = params[16:20]
        W = [w if l != SKIP else w / math.sqrt(2.0) for l, w in enumerate(W)]   # cat([h, e]) / sqrt(2) folded
        x3 = pts * 3.0
        e, dc, d2c, dim = _enc_parts(x3)
        xs, s1 = [], []
        x = e
        for l in range(N_LAYERS):
            if l == SKIP:
                x = torch.cat([x, e], dim=1)
            xs.append(x)
            z = F.linear(x, W[l], b[l])
            t = z * 100.0
            ez = torch.exp(torch.clamp(t, max=80.0))
            s1.append(torch.where(t > 20.0, torch.ones_like(t), ez / (ez + 1.0)))
            x = F.softplus(z, beta=100)
        h7 = x
        sdf = F.linear(h7, ws, bs) / 3.0
        feat = F.linear(h7, Wf, bf)
        # reverse chain for g
        a_next = (ws / 3.0).expand(pts.shape[0], -1)        # a_8
        a_list = [None] * (N_LAYERS + 1)
        a_list[N_LAYERS] = a_next
        ge_skip = None
        for l in range(N_LAYERS - 1, -1, -1):
            a = (s1[l] * a_next) @ W[l]                       # a_l: gradient w.r.t. x_l
            if l == SKIP:
                ge_skip = a[:, 217:]
                a_next = a[:, :217]
            else:
                a_next = a
            a_list[l] = a_next                                # gradient w.r.t. h_{l-1} (or the embedding for l = 0)
        ge = a_list[0] + ge_skip
        g = 3.0 * _scatter_dims(ge * dc, dim)
        ctx.save_for_backward(pts, *W, ws, Wf, *xs, *s1, *[a_list[l] for l in range(1, N_LAYERS + 1)], ge, dc, d2c, h7)
        ctx.dim = dim
        return sdf, feat, g

    @staticmethod
    def backward(ctx, sbar, fbar, gbar):
        sv = ctx.saved_tensors
        pts = sv[0]
        W = list(sv[1:9])
        ws, Wf = sv[9], sv[10]
        xs = list(sv[11:19])
        s1 = list(sv[19:27])
        a_up = list(sv[27:35])          # a_up[l] = a_{l+1} restricted to h_l's width, l = 0..7
        ge, dc, d2c, h7 = sv[35:39]
        dim = ctx.dim
        P = pts.shape[0]
        sbar = torch.zeros(P, 1, dtype=pts.dtype, device=pts.device) if sbar is None else sbar
        fbar = torch.zeros(P, 256, dtype=pts.dtype, device=pts.device) if fbar is None else fbar
        gbar = torch.zeros(P, 3, dtype=pts.dtype, device=pts.device) if gbar is None else gbar

        # ---- adjoint of g = 3 * sum_e ge[e] dc[e]: tangent adjoints run FORWARD through the layers ----
        gb_e = gbar[:, dim]                                   # gbar of the coordinate each entry depends on
        ge_bar = 3.0 * dc * gb_e                              # [P,39]
        p3_bar = 3.0 * _scatter_dims(ge * d2c * gb_e, dim)    # through the encoding's second derivative
        dW = [None] * N_LAYERS
        coup = [None] * N_LAYERS
        abar = ge_bar
        ws_bar = torch.zeros_like(ws)
        for l in range(N_LAYERS):
            if l == SKIP:
                abar = torch.cat([abar, ge_bar], dim=1)       # adjoint of a_4 = [a_4h, skip part]
            tbar = abar @ W[l].t()                            # [P,out_l]
            t_l = s1[l] * a_up[l]
            dW[l] = t_l.t() @ abar                            # term 2 of dW_l
            coup[l] = (100.0 * s1[l] * (1.0 - s1[l])) * a_up[l] * tbar
            if l == N_LAYERS - 1:
                ws_bar = ws_bar + (s1[l] * tbar).sum(0, keepdim=True) / 3.0
            else:
                abar = s1[l] * tbar                           # adjoint of a_{l+1} (h_l-wide)
        # ---- value adjoints run in REVERSE ----
        hbar = fbar @ Wf + sbar * (ws / 3.0)
        Wf_bar = fbar.t() @ h7
        bf_bar = fbar.sum(0)
        ws_bar = ws_bar + (sbar * h7).sum(0, keepdim=True) / 3.0
        bs_bar = sbar.sum(0).reshape(-1) / 3.0
        db = [None] * N_LAYERS
        e_skip_bar = None
        for l in range(N_LAYERS - 1, -1, -1):
            zbar = s1[l] * hbar + coup[l]
            dW[l] = dW[l] + zbar.t() @ xs[l]
            db[l] = zbar.sum(0)
            xbar = zbar @ W[l]
            if l == SKIP:
                e_skip_bar = xbar[:, 217:]
                hbar = xbar[:, :217]
            else:
                hbar = xbar
        e_bar = hbar + e_skip_bar
        p3_bar = p3_bar + _scatter_dims(e_bar * dc, dim)
        dW[SKIP] = dW[SKIP] / math.sqrt(2.0)                  # back to the unscaled W4
        return (p3_bar * 3.0, *dW, *db, ws_bar, bs_bar, Wf_bar, bf_bar)



def sdf_value_feat_grad_manual(dense: Dict[str, torch.Tensor], pts: torch.Tensor):
    args = [dense[f"sdf_w{l}"] for l in range(8)] + [dense[f"sdf_b{l}"] for l in range(8)] + \
           [dense["sdf_head_w"], dense["sdf_head_b"], dense["feat_w"], dense["feat_b"]]
    return SdfValueFeatGrad.apply(pts, *args)


def _sdf_net(d: Dict[str, torch.Tensor], pts: torch.Tensor):
    e = _enc(pts * 3.0, 6)
    h = e
    for l in range(8):
        if l == 4:
            h = torch.cat([h, e], dim=1) / math.sqrt(2.0)
        h = F.softplus(F.linear(h, d[f"sdf_w{l}"], d[f"sdf_b{l}"]), beta=100)
    return F.linear(h, d["sdf_head_w"], d["sdf_head_b"]) / 3.0, F.linear(h, d["feat_w"], d["feat_b"])


def _color_net_torch(d, feat, pts, normal, per_ray, n, T, hints):
    """Reflectance net in torch ops, layer 0 by column blocks of the reference's 361-wide input
      [pts 0:3 | enc(view) 3:30 | normal 30:33 | enc(pl) 33:60 | feat 60:316 | enc(vis) 316:325 | enc(cue) 325:361]
    (fields/reflectance_network.py:77-82): the view / light / visibility / cue encodings are constant along a ray, so
    their contribution is one [N,99] x [99,256] product broadcast over the 128 samples instead of a 361-wide
    concatenation per sample; autograd carries the ray gradients through the small per-ray part."""
    w0, b0 = d["col_w0"], d["col_b0"]
    ray_cols, pn_cols = _col_index(w0.device, hints)
    x = _linear(feat, w0[:, 60:316], b0)                                                   # [P,256] the big block
    x = x + _linear(torch.cat([pts, normal], dim=-1), w0[:, pn_cols], torch.zeros_like(b0))  # per-sample 6 columns
    x = (x.reshape(n, T, -1) + (torch.cat(per_ray, dim=-1) @ w0[:, ray_cols].t())[:, None, :]).reshape(n * T, -1)
    x = torch.relu(x)
    for l in range(1, 5):
        x = _linear(x, d[f"col_w{l}"], d[f"col_b{l}"])
        if l < 4:
            x = torch.relu(x)
    return torch.sigmoid(x).reshape(n, T, 3)



def render_core_torch(d: Dict[str, torch.Tensor], variance: torch.Tensor, o, dirs, pl, mid_z, dists, vis, cue, cos_anneal: float,
                      background_rgb, analytic_normal: bool = False, packed=None, pre=None, dyn=None, hint_grad=None,
                      n_real: int = 128, sdf_impl: str = "manual"):
    """autograd_core.render_core in torch ops: ``sdf_impl`` "manual" (hand-derived sweeps) | "autograd" (second-order graph)."""
    assert hint_grad is None and n_real == 128, "the torch back-ends cover the default (hint-constant, 128-sample) training path"
    n, T = mid_z.shape
    pts = (o[:, None, :] + dirs[:, None, :] * mid_z[..., None]).reshape(-1, 3)
    if sdf_impl == "autograd":
        if not pts.requires_grad:
            pts.requires_grad_(True)
        sdf, feat = _sdf_net(d, pts)
        (grad,) = torch.autograd.grad(sdf, pts, torch.ones_like(sdf), create_graph=True, retain_graph=True)
    else:
        sdf, feat, grad = sdf_value_feat_grad_manual(d, pts)
    inv_s = torch.exp(variance * 10.0).clip(1e-6, 1e6)
    view = dirs[:, None, :].expand(n, T, 3).reshape(-1, 3)
    true_cos = (view * grad).sum(-1, keepdim=True)
    iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal) + F.relu(-true_cos) * cos_anneal)
    dd = dists.reshape(-1, 1)
    c_prev = torch.sigmoid((sdf - iter_cos * dd * 0.5) * inv_s)
    c_next = torch.sigmoid((sdf + iter_cos * dd * 0.5) * inv_s)
    alpha = ((c_prev - c_next + 1e-5) / (c_prev + 1e-5)).clip(0.0, 1.0).reshape(n, T)
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-7], dim=-1), dim=-1)[:, :-1]
    weights = alpha * trans
    n_hat = F.normalize(grad, dim=-1)
    per_ray = [_enc(dirs, 4), _enc(pl, 4)]
    if vis is not None:
        per_ray += [_enc(vis, 4), _enc(cue, 4)]
    normal = grad if analytic_normal else n_hat
    col = _color_net_torch(d, feat, pts, normal, per_ray, n, T, vis is not None)
    rgb = (col * weights[..., None]).sum(1)
    if background_rgb is not None:
        rgb = rgb + background_rgb * (1.0 - weights.sum(-1, keepdim=True))
    return dict(rgb=rgb, weights=weights, analytic_normals=grad.reshape(n, T, 3),
                normalized_analytic_normals=n_hat.reshape(n, T, 3), s_val=(1.0 / inv_s).expand(n, T))


@contextlib.contextmanager
def use_torch_backend(model, sdf_impl: str):
    """Run ``model``'s training forward with the torch formulation of render_core (the graph-less stages stay in HIP)."""
    real, fused = autograd_core.render_core, model.max_fused_train_rays
    autograd_core.render_core = lambda *a, **k: render_core_torch(*a, **k, sdf_impl=sdf_impl)
    model.max_fused_train_rays = 0          # the fused training call feeds the HIP sweeps; the torch back-ends evaluate themselves
    try:
        yield
    finally:
        autograd_core.render_core, model.max_fused_train_rays = real, fused
