"""The outside-NeRF background (``renderer.use_outside_nerf``; models/neus_hint_model.py:260-266, :434-473, :516-519, :630-633,
:677-724; fields/nerf_density_field.py) - off in every configuration the reference ships.

Division of labour: everything NeuS - both hierarchical samplers, the SDF network and its gradient, the alpha stage INCLUDING the
NeuS / NeRF blend and the transmittance handed to the samples beyond the sphere, hit point, shadow march, cue, the reflectance
network, and all their adjoints - stays in the HIP kernels (``nrh_sample_primary``, ``nrh_render_forward`` / ``_train`` with
``NrhNet.bg_alpha``, ``nrh_alpha_blend_*``).  The background network itself is an 8 x 256 ReLU MLP with no structure to exploit
beyond its GEMMs: it runs here as library GEMMs (``F.linear``) on the GPU, with autograd for its backward, as do the 32-sample
tail composite and the colour blend (elementwise work on [N,160] arrays).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from .autograd_core import _enc

N_OUTSIDE = 32


class OutsideNeRF(nn.Module):
    """Parameter container + forward of the reference's ``NeRF`` (fields/nerf_density_field.py:30-89) with its state-dict names:
    pts_linears.0..7, views_linears.0, feature_linear, alpha_linear, rgb_linear (constructed in the reference's order, so the
    init RNG stream of a model built with the background matches too)."""

    def __init__(self, d_in: int = 4, d_in_view: int = 6, d_hidden: int = 256, n_layers: int = 8, multi_res: int = 10,
                 multi_res_view: int = 4, skips=(4,)):
        super().__init__()
        self.multi_res, self.multi_res_view, self.skips = multi_res, multi_res_view, tuple(skips)
        ch, ch_view = d_in * (2 * multi_res + 1), d_in_view * (2 * multi_res_view + 1)
        self.pts_linears = nn.ModuleList([nn.Linear(ch, d_hidden)] + [
            nn.Linear(d_hidden + ch, d_hidden) if i in self.skips else nn.Linear(d_hidden, d_hidden) for i in range(n_layers - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(ch_view + d_hidden, d_hidden // 2)])
        self.feature_linear = nn.Linear(d_hidden, d_hidden)
        self.alpha_linear = nn.Linear(d_hidden, 1)
        self.rgb_linear = nn.Linear(d_hidden // 2, 3)

    def forward(self, pts4: torch.Tensor, views: torch.Tensor, pls: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (density [P,1], rgb before the sigmoid [P,3])  (:66-89)."""
        x = _enc(pts4, self.multi_res)
        v = _enc(torch.cat([views, pls], dim=-1), self.multi_res_view)
        h = x
        for i, lin in enumerate(self.pts_linears):
            h = F.relu(lin(h))
            if i in self.skips:
                h = torch.cat([x, h], dim=-1)
        density = self.alpha_linear(h)
        h = F.relu(self.views_linears[0](torch.cat([self.feature_linear(h), v], dim=-1)))
        return density, self.rgb_linear(h)


def outside_z(far: torch.Tensor, n_samples: int, t_rand=None) -> torch.Tensor:
    """Sample positions beyond the unit sphere, inverse-depth spaced (models/neus_hint_model.py:677-693): far [N,1] -> [N,32]."""
    u = torch.linspace(1e-3, 1.0 - 1.0 / (N_OUTSIDE + 1.0), N_OUTSIDE).to(far)     # linspace on the host, as the reference's default device does
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
    density, col = nerf(pts4, d[:, None, :].expand(n, m, 3).reshape(-1, 3), pl[:, None, :].expand(n, m, 3).reshape(-1, 3))
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


def composite(weights128, tail_t, inside, color128, bg_alpha, bg_col, background_rgb) -> Dict[str, torch.Tensor]:
    """What render_core does after the alpha stage when a background is present (:520-523, :630-637): the 32 samples beyond the
    sphere continue the transmittance product from ``tail_t``; a sample outside the unit sphere shows the background's colour."""
    a_tail = bg_alpha[:, 128:]
    trans = torch.cumprod(torch.cat([torch.ones_like(a_tail[:, :1]), 1.0 - a_tail + 1e-7], dim=-1), dim=-1)[:, :-1]
    w_tail = a_tail * tail_t * trans
    weights = torch.cat([weights128, w_tail], dim=-1)
    ins = inside[..., None]
    col = torch.cat([color128 * ins + bg_col[:, :128] * (1.0 - ins), bg_col[:, 128:]], dim=1)
    rgb = (col * weights[..., None]).sum(1)
    if background_rgb is not None:
        rgb = rgb + background_rgb * (1.0 - weights.sum(-1, keepdim=True))
    return dict(rgb=rgb, weights=weights)
