"""The outside-NeRF background (``renderer.use_outside_nerf``; models/neus_hint_model.py:260-266, :434-473, :516-519, :630-633,
:677-724; fields/nerf_density_field.py) - off in every configuration the reference ships.

Division of labour: everything NeuS - both hierarchical samplers, the SDF network and its gradient, the alpha stage INCLUDING the
NeuS / NeRF blend and the transmittance handed to the samples beyond the sphere, hit point, shadow march, cue, the reflectance
network, and all their adjoints - stays in the HIP kernels (``nrh_sample_primary``, ``nrh_render_forward`` / ``_train`` with
``NrhNet.bg_alpha``, ``nrh_alpha_blend_*``).  The background network itself - an 8 x 256 ReLU MLP with a skip connection, a density
head and a 128-wide view branch - is the kernel pair of csrc/nrh_outside.hip (``nrh_outside_forward`` / ``nrh_outside_backward``, the
transposed register chain of the other per-point networks, both precisions) and its weight gradients are jobs of ``nrh_dw_gemm``;
there is no library GEMM and no autograd inside the network.  What stays in torch is elementwise work on [N,160] arrays: the sample
positions, alpha = 1 - exp(-softplus(density) dist), the 32-sample tail composite, and the chain rule through the two positional
encodings for the ray gradients.
"""
from __future__ import annotations

import ctypes
import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib, dw
from .autograd_core import _freqs
from .packing import pack_stage, pack_stage_h3

N_OUTSIDE = 32
X_COLS, V_COLS, X_REAL, V_REAL = 96, 64, 84, 54      # csrc/nrh_outside.hip: ON_X, ON_V, ON_XREAL, ON_VREAL


class _Affine(nn.Module):
    """``weight`` [out, in] and ``bias`` [out] of one layer, initialised with nn.Linear's arithmetic (kaiming_uniform_(a = sqrt 5),
    then bias ~ U(+-1 / sqrt(fan_in)): the same draws in the same order, so a model built under a seed equals the reference's).
    A parameter container only: the layers are evaluated by the HIP kernels."""

    def __init__(self, fan_in: int, fan_out: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fan_out, fan_in))
        self.bias = nn.Parameter(torch.empty(fan_out))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
        nn.init.uniform_(self.bias, -bound, bound)


class OutsideNeRF(nn.Module):
    """The reference's ``NeRF`` (fields/nerf_density_field.py:30-89) with its state-dict names: pts_linears.0..7, views_linears.0,
    feature_linear, alpha_linear, rgb_linear (constructed in the reference's order, so the init RNG stream of a model built with
    the background matches too).  ``forward`` runs csrc/nrh_outside.hip (GPU tensors only)."""

    precision = "f16x3"      # set by the renderer that owns the module

    def __init__(self, d_in: int = 4, d_in_view: int = 6, d_hidden: int = 256, n_layers: int = 8, multi_res: int = 10,
                 multi_res_view: int = 4, skips=(4,)):
        super().__init__()
        if (d_in, d_in_view, d_hidden, n_layers, multi_res, multi_res_view, tuple(skips)) != (4, 6, 256, 8, 10, 4, (4,)):
            raise ValueError("the outside-NeRF kernels are built for the default 8 x 256 / multires 10 + 4 / skips = [4] network")
        self.multi_res, self.multi_res_view, self.skips = multi_res, multi_res_view, tuple(skips)
        ch, ch_view = d_in * (2 * multi_res + 1), d_in_view * (2 * multi_res_view + 1)
        self.pts_linears = nn.ModuleList([_Affine(ch, d_hidden)] + [
            _Affine(d_hidden + ch, d_hidden) if i in self.skips else _Affine(d_hidden, d_hidden) for i in range(n_layers - 1)])
        self.views_linears = nn.ModuleList([_Affine(ch_view + d_hidden, d_hidden // 2)])
        self.feature_linear = _Affine(d_hidden, d_hidden)
        self.alpha_linear = _Affine(d_hidden, 1)
        self.rgb_linear = _Affine(d_hidden // 2, 3)

    def ordered_parameters(self) -> List[nn.Parameter]:
        """weight, bias of pts_linears 0..7, views_linears.0, feature_linear, alpha_linear, rgb_linear (24 tensors)"""
        mods = list(self.pts_linears) + [self.views_linears[0], self.feature_linear, self.alpha_linear, self.rgb_linear]
        return [t for m in mods for t in (m.weight, m.bias)]

    def forward(self, pts4: torch.Tensor, views: torch.Tensor, pls: torch.Tensor, pts_per_ray: int = 1) -> Tuple[torch.Tensor, torch.Tensor]:
        """pts4 [P,4]; views, pls [P / pts_per_ray, 3] -> (density [P,1], rgb before the sigmoid [P,3])  (:66-89)."""
        return OutsideNetHip.apply(pts4, views, pls, _lib.PRECISIONS[self.precision], int(pts_per_ray), *self.ordered_parameters())


def pack_outside(params: List[torch.Tensor], precision: int, transposed: bool):
    """The network's 24 tensors (OutsideNeRF.ordered_parameters) -> the kernels' buffers (csrc/nrh_outside.hip):
    forward stream N0 | N1..N4 | N5a (h part) | N5b (x part) | N6 | N7 | HF (feature + density row) | VA | VB | RGB and the bias
    table, or (``transposed``) the adjoint sweep's stream TRGB | TVA | TVB | THF | T7 | T6 | T5x | T5h | T4..T1 | T0."""
    ps = pack_stage if precision == 0 else pack_stage_h3
    w = [p.detach().float() for p in params[0::2]]
    b = [p.detach().float() for p in params[1::2]]
    W, Wv, Wf, Wa, Wrgb = w[:8], w[8], w[9], w[10], w[11]
    if not transposed:
        parts = [ps(W[0], 256, X_COLS)] + [ps(W[l], 256, 256) for l in (1, 2, 3, 4)] + \
                [ps(W[5][:, X_REAL:], 256, 256), ps(W[5][:, :X_REAL], 256, X_COLS), ps(W[6], 256, 256), ps(W[7], 256, 256),
                 ps(torch.cat([Wf, Wa], dim=0), 288, 256), ps(Wv[:, :256], 128, 256), ps(Wv[:, 256:], 128, V_COLS), ps(Wrgb, 32, 128)]
        pad = lambda v, n: torch.cat([v, v.new_zeros(n - v.numel())])
        bias = torch.cat([torch.cat(b[:8]), b[9], pad(b[10], 16), b[8], pad(b[11], 32)])
        return torch.cat(parts).contiguous(), bias.contiguous()
    parts = [ps(Wrgb.t(), 128, 32), ps(Wv[:, :256].t(), 256, 128), ps(Wv[:, 256:].t(), V_COLS, 128), ps(Wf.t(), 256, 256),
             ps(W[7].t(), 256, 256), ps(W[6].t(), 256, 256), ps(W[5][:, :X_REAL].t(), X_COLS, 256), ps(W[5][:, X_REAL:].t(), 256, 256)] + \
            [ps(W[l].t(), 256, 256) for l in (4, 3, 2, 1)] + [ps(W[0].t(), X_COLS, 256)]
    return torch.cat(parts).contiguous()


def _enc_adjoint(x: torch.Tensor, n_freq: int, ebar: torch.Tensor) -> torch.Tensor:
    """Adjoint of the NeRF encoding [x, sin(x_d 2^k), sin(x_d 2^k + pi/2)] (fields/encodings.py:168-174): ebar [P, D (2 F + 1)] -> [P, D]."""
    D = x.shape[-1]
    fr = _freqs(n_freq, x)
    s = x[..., None] * fr                                                   # [P, D, F]
    gs = ebar[:, D:D + D * n_freq].reshape(-1, D, n_freq)
    gc = ebar[:, D + D * n_freq:D + 2 * D * n_freq].reshape(-1, D, n_freq)
    return ebar[:, :D] + ((gs * torch.cos(s) + gc * torch.cos(s + math.pi / 2.0)) * fr).sum(-1)


_MAPS: Dict[tuple, torch.Tensor] = {}


def _arange_i32(lo: int, n: int, device) -> torch.Tensor:
    key = (lo, n, str(device))
    if key not in _MAPS:
        _MAPS[key] = torch.arange(lo, lo + n, dtype=torch.int32, device=device)
    return _MAPS[key]


class OutsideNetHip(torch.autograd.Function):
    """(pts4 [P,4], views [N,3], pls [N,3], 24 parameters) -> (density [P,1], rgb before the sigmoid [P,3]) through
    nrh_outside_forward; the backward is nrh_outside_backward + 13 jobs of nrh_dw_gemm (every weight and bias gradient) + the
    chain rule through the two encodings in torch (elementwise)."""

    @staticmethod
    def forward(ctx, pts4, views, pls, precision: int, pts_per_ray: int, *params):
        if not pts4.is_cuda:
            raise RuntimeError("the outside-NeRF network runs on the GPU only (csrc/nrh_outside.hip); the CPU restatement is oracle.neus_oracle.nerf_forward")
        lib, P = _lib.load(), _lib.ptr
        dev, npts = pts4.device, pts4.shape[0]
        f32c = lambda t: t.detach().to(torch.float32).contiguous()
        x4, vw, pl = f32c(pts4), f32c(views), f32c(pls)
        if npts % pts_per_ray != 0 or vw.shape[0] * pts_per_ray != npts or pl.shape != vw.shape:
            raise ValueError("views / pls must hold one row per group of pts_per_ray points")
        train = any(ctx.needs_input_grad)
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        dens, rgb = new(npts, 1), new(npts, 3)
        sv = None
        with torch.cuda.device(dev):
            w, b = pack_outside(list(params), precision, transposed=False)
            if train:
                if npts % 16 != 0:
                    raise ValueError("training the outside-NeRF network needs a multiple of 16 points")
                sv = dict(x=new(npts, X_COLS), v=new(npts, V_COLS), h=new(8, npts, 256), f=new(npts, 256), hv=new(npts, 128))
            sp = (lambda k: P(sv[k])) if train else (lambda k: None)
            _lib.check(lib.nrh_outside_forward(precision, P(w, w.dtype), P(b), P(x4), P(vw), P(pl), pts_per_ray, npts, P(dens), P(rgb),
                                               sp("x"), sp("v"), sp("h"), sp("f"), sp("hv"), _lib.stream_handle()), "nrh_outside_forward")
        if train:
            ctx.save_for_backward(x4, vw, pl, sv["x"], sv["v"], sv["h"], sv["f"], sv["hv"], *[p.detach() for p in params])
            ctx.cfg = (precision, pts_per_ray)
        return dens, rgb

    @staticmethod
    def backward(ctx, dbar, cbar):
        lib, P = _lib.load(), _lib.ptr
        x4, vw, pl, sx, svv, sh, sf, shv = ctx.saved_tensors[:8]
        params = list(ctx.saved_tensors[8:])
        precision, ppr = ctx.cfg
        dev, npts = x4.device, x4.shape[0]
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        zeros = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        dbar = zeros(npts, 1) if dbar is None else dbar.to(torch.float32).contiguous()
        cbar = zeros(npts, 3) if cbar is None else cbar.to(torch.float32).contiguous()
        zbar, fbar, zvbar, xbar, vbar = new(8, npts, 256), new(npts, 256), new(npts, 128), new(npts, X_COLS), new(npts, V_COLS)
        with torch.cuda.device(dev):
            wt = pack_outside(params, precision, transposed=True)
            walpha = params[20].detach().float().reshape(-1).contiguous()
            _lib.check(lib.nrh_outside_backward(precision, P(wt, wt.dtype), P(walpha), P(dbar), P(cbar), P(sh), P(shv), npts, P(zbar), P(fbar),
                                                P(zvbar), P(xbar), P(vbar), _lib.adjoint_scale_from_seeds((dbar, cbar), npts // max(1, ppr)), _lib.stream_handle()),
                       "nrh_outside_backward")
            g = [torch.empty_like(p, dtype=torch.float32) for p in params]      # weight, bias pairs in parameter order
            J = dw.Job
            jobs = [J([zbar[0]], [sx], 256, X_REAL, g[0], colsum_a=g[1])]
            for l in (1, 2, 3, 4, 6, 7):
                jobs.append(J([zbar[l]], [sh[l - 1]], 256, 256, g[2 * l], colsum_a=g[2 * l + 1]))
            jobs += [J([zbar[5]], [sx], 256, X_REAL, g[10], col_map=_arange_i32(0, X_REAL, dev), colsum_a=g[11]),
                     J([zbar[5]], [sh[4]], 256, 256, g[10], col_map=_arange_i32(X_REAL, 256, dev)),
                     J([zvbar], [sf], 128, 256, g[16], col_map=_arange_i32(0, 256, dev), colsum_a=g[17]),
                     J([zvbar], [svv], 128, V_REAL, g[16], col_map=_arange_i32(256, V_REAL, dev)),
                     J([fbar], [sh[7]], 256, 256, g[18], colsum_a=g[19]),
                     J([sh[7]], [dbar.reshape(npts, 1)], 256, 1, g[20], transpose=True, colsum_b=g[21]),
                     J([shv], [cbar], 128, 3, g[22], transpose=True, colsum_b=g[23])]
            dw.run(jobs, npts)
        need = ctx.needs_input_grad
        g_pts = _enc_adjoint(x4, 10, xbar) if need[0] else None
        g_v = g_pl = None
        if need[1] or need[2]:
            v6 = torch.cat([vw, pl], dim=-1)[:, None, :].expand(-1, ppr, -1).reshape(npts, 6)
            gv6 = _enc_adjoint(v6, 4, vbar).reshape(-1, ppr, 6).sum(1)
            g_v, g_pl = (gv6[:, :3] if need[1] else None), (gv6[:, 3:] if need[2] else None)
        return (g_pts, g_v, g_pl, None, None, *g)


def outside_z(far: torch.Tensor, n_samples: int, t_rand=None) -> torch.Tensor:
    """Sample positions beyond the unit sphere, inverse-depth spaced (models/neus_hint_model.py:677-693): far [N,1] -> [N,32]."""
    key = ("outside_u", str(far.device), far.dtype)
    if key not in _MAPS:      # linspace on the host, as the reference's default device does; cached per device (a captured step may not copy from the host)
        _MAPS[key] = torch.linspace(1e-3, 1.0 - 1.0 / (N_OUTSIDE + 1.0), N_OUTSIDE).to(far)
    u = _MAPS[key]
    if t_rand is not None:      # stratified jitter in training
        mids = 0.5 * (u[1:] + u[:-1])
        upper, lower = torch.cat([mids, u[-1:]]), torch.cat([u[:1], mids])
        u = lower[None, :] + (upper - lower)[None, :] * t_rand
    return far / torch.flip(u, dims=[-1]) + 1.0 / n_samples


def render_outside(nerf: OutsideNeRF, o, d, pl, z, sample_dist: float):
    """``render_outside`` (:434-473) at the sorted positions z [N,160]: section mid-points in the inverted-sphere parameterisation
    (p / |p|, 1 / |p|), alpha = 1 - exp(-softplus(density) dist).  -> (alpha [N,160], colour [N,160,3])."""
    n, m = z.shape
    dists = torch.cat([z[:, 1:] - z[:, :-1], z.new_full((n, 1), sample_dist)], dim=-1)
    mid = z + dists * 0.5
    pts = o[:, None, :] + d[:, None, :] * mid[..., None]
    r = torch.linalg.norm(pts, ord=2, dim=-1, keepdim=True).clip(1.0, 1e10)
    pts4 = torch.cat([pts / r, 1.0 / r], dim=-1).reshape(-1, 4)
    density, col = nerf(pts4, d, pl, pts_per_ray=m)      # view direction and light position are per ray (:452-456)
    alpha = 1.0 - torch.exp(-F.softplus(density.reshape(n, m)) * dists)
    return alpha, torch.sigmoid(col).reshape(n, m, 3)


class AlphaBlendHip(torch.autograd.Function):
    """AlphaWeightsNormalsHip with the background blend (:516-519): alpha <- alpha inside + bg_alpha[:, :128] (1 - inside);
    (sdf, grad, dirs, dists, inside [N,128], bg_alpha [N,160], variance) -> (weights [N,128], n_hat [P,3], tail_t [N,1] =
    transmittance behind sample 127).  Forward and adjoint are the alpha-stage kernel pair (nrh_alpha_blend_*)."""

    @staticmethod
    def forward(ctx, sdf, grad, dirs, dists, inside, bg_alpha, variance, inv_s: float, cos_anneal: float, dyn=None):
        lib = _lib.load()
        n = dirs.shape[0]
        f32c = lambda t: t.detach().to(torch.float32).contiguous()
        sdf_c, grad_c, dirs_c, dists_c, ins_c, bg_c = f32c(sdf), f32c(grad), f32c(dirs), f32c(dists), f32c(inside), f32c(bg_alpha)
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dirs.device)
        weights, nhat, tail = new(n, 128), new(n * 128, 3), new(n, 1)
        P = _lib.ptr
        _lib.check(lib.nrh_alpha_blend_forward(P(sdf_c), P(grad_c), P(dirs_c), P(dists_c), P(ins_c), P(bg_c), float(inv_s), float(cos_anneal),
                                               P(dyn), n, P(weights), P(nhat), P(tail), _lib.stream_handle()), "nrh_alpha_blend_forward")
        ctx.save_for_backward(sdf_c, grad_c, dirs_c, dists_c, ins_c, bg_c)
        ctx.consts, ctx.dyn = (float(inv_s), float(cos_anneal)), dyn
        return weights, nhat, tail

    @staticmethod
    def backward(ctx, wbar, nbar, tbar):
        lib = _lib.load()
        sdf_c, grad_c, dirs_c, dists_c, ins_c, bg_c = ctx.saved_tensors
        inv_s, cos_anneal = ctx.consts
        n, dev = dirs_c.shape[0], dirs_c.device
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        zeros = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        wbar = zeros(n, 128) if wbar is None else wbar.to(torch.float32).contiguous()
        tbar = zeros(n, 1) if tbar is None else tbar.to(torch.float32).contiguous()
        nbar = None if nbar is None else nbar.to(torch.float32).contiguous()
        sdf_bar, grad_bar, rd_bar, invs_bar, bg_bar = new(n * 128, 1), new(n * 128, 3), new(n, 3), new(n), zeros(n, 160)
        bg128 = new(n, 128)
        P = _lib.ptr
        _lib.check(lib.nrh_alpha_blend_backward(P(sdf_c), P(grad_c), P(dirs_c), P(dists_c), P(ins_c), P(bg_c), inv_s, cos_anneal, P(ctx.dyn), n,
                                                P(wbar), P(nbar), P(tbar), P(sdf_bar), P(grad_bar), P(rd_bar), P(invs_bar), P(bg128),
                                                _lib.stream_handle()), "nrh_alpha_blend_backward")
        bg_bar[:, :128] = bg128
        if ctx.dyn is not None:
            s_dev = ctx.dyn[0]
            var_bar = invs_bar.sum() * torch.where((s_dev > 1e-6) & (s_dev < 1e6), 10.0 * s_dev, torch.zeros_like(s_dev))
        else:
            var_bar = invs_bar.sum() * (10.0 * inv_s if 1e-6 < inv_s < 1e6 else 0.0)
        return sdf_bar, grad_bar, rd_bar, None, None, bg_bar, var_bar, None, None, None


class _ExclusiveCumprod(torch.autograd.Function):
    """T_k = prod_{i<k} x_i along the last dimension, the forward exactly as render_core writes it (cumprod of [1, x][:-1], :520-523).
    Its own backward because torch.cumprod's tests the input for zeros on the host - a synchronisation a captured training step
    cannot contain; here x = 1 - alpha + 1e-7 > 0, and the zero-free formula is the one torch takes then: rev-cumsum(g y) / x."""

    @staticmethod
    def forward(ctx, x):
        y = torch.cumprod(torch.cat([torch.ones_like(x[:, :1]), x], dim=-1), dim=-1)[:, :-1]
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        gy = g * y
        return (gy.flip(-1).cumsum(-1).flip(-1) - gy) / x


def composite(weights128, tail_t, inside, color128, bg_alpha, bg_col, background_rgb) -> Dict[str, torch.Tensor]:
    """What render_core does after the alpha stage when a background is present (:520-523, :630-637): the 32 samples beyond the
    sphere continue the transmittance product from ``tail_t``; a sample outside the unit sphere shows the background's colour."""
    a_tail = bg_alpha[:, 128:]
    trans = _ExclusiveCumprod.apply(1.0 - a_tail + 1e-7)
    w_tail = a_tail * tail_t * trans
    weights = torch.cat([weights128, w_tail], dim=-1)
    ins = inside[..., None]
    col = torch.cat([color128 * ins + bg_col[:, :128] * (1.0 - ins), bg_col[:, 128:]], dim=1)
    rgb = (col * weights[..., None]).sum(1)
    if background_rgb is not None:
        rgb = rgb + background_rgb * (1.0 - weights.sum(-1, keepdim=True))
    return dict(rgb=rgb, weights=weights)
