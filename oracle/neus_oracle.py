"""CPU ORACLE for the NRHints volumetric-rendering hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain PyTorch-CPU tensor arithmetic, what the reference's
``NeuSHintRenderer.forward`` computes.  It is the checker the HIP
path is compared against and the timed leg of ``bench.py``'s ``cpu_baseline``.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg
may import it; nothing under ``nrhints_amd/`` does.  It is not a fallback.

Parity pin: every function below is checked in ``tests/test_oracle_golden.py``
against fixtures produced by importing the reference itself in the build
container (``tests/golden/make_golden.py``); the reference ships no tests or
golden vectors of its own.

Two evaluation strategies, same mathematics:

* ``mode="as_written"`` - the reference's call pattern: 13 full SDF-net forwards
  per render (feature head always evaluated), d(sdf)/dx by autograd.  This is
  what the CPU baseline times.
* ``mode="minimal"``   - one SDF evaluation per point, analytic reverse chain for
  the gradient (what the HIP kernels do).  Used to show the restructuring is
  value-preserving.

All citations are ``path:line`` under /root/reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

SPEC_ROUGHNESS = (0.02, 0.05, 0.13, 0.34)  # models/neus_hint_model.py:161


# --------------------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------------------
def fold_weight_norm(g: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """W = g * v / ||v||_2 per output row (nn.utils.weight_norm, dim=0; fields/sdf_field.py:81-82)."""
    return v * (g / v.norm(dim=1, keepdim=True))


@dataclass
class OracleParams:
    sdf_w: List[torch.Tensor]      # 8 trunk layers [out,in]
    sdf_b: List[torch.Tensor]
    sdf_head_w: torch.Tensor       # [1,256]
    sdf_head_b: torch.Tensor
    feat_w: torch.Tensor           # [256,256]
    feat_b: torch.Tensor
    col_w: List[torch.Tensor]      # 5 layers
    col_b: List[torch.Tensor]
    variance: torch.Tensor         # scalar


def params_from_state(state, dtype=torch.float32) -> OracleParams:
    """Dense (weight-norm folded) parameters from a reference-layout state dict (SURVEY.md §5 key contract).
    Values may be numpy arrays or torch tensors; tensors that require grad keep their autograd history, so
    gradients w.r.t. the raw weight_g / weight_v / bias / variance parameters can be taken through the fold."""
    t = {k: (v.to(dtype) if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v)).to(dtype))
         for k, v in state.items()}

    def lin(prefix):
        return fold_weight_norm(t[prefix + ".weight_g"], t[prefix + ".weight_v"]), t[prefix + ".bias"]

    sw, sb = zip(*[lin(f"sdf_network.lin{i}") for i in range(8)])
    hw, hb = lin("sdf_network.out_sdf")
    fw, fb = lin("sdf_network.out_feat")
    cw, cb = zip(*[lin(f"color_network.lin{i}") for i in range(5)])
    return OracleParams(list(sw), list(sb), hw, hb, fw, fb, list(cw), list(cb), t["deviation_network.variance"])


# --------------------------------------------------------------------------------------
# fields
# --------------------------------------------------------------------------------------
def nerf_encode(x: torch.Tensor, n_freq: int) -> torch.Tensor:
    """[x, sin(x_d 2^k) (d-major, k-minor), sin(x_d 2^k + pi/2)]   (fields/encodings.py:155-176)."""
    freqs = 2.0 ** torch.linspace(0.0, n_freq - 1, n_freq, dtype=x.dtype, device=x.device)
    s = (x[..., None] * freqs).reshape(*x.shape[:-1], -1)
    return torch.cat([x, torch.sin(torch.cat([s, s + math.pi / 2.0], dim=-1))], dim=-1)


def softplus100(x: torch.Tensor) -> torch.Tensor:
    """nn.Softplus(beta=100), threshold 20 (fields/sdf_field.py:104)."""
    return F.softplus(x, beta=100)


def sdf_forward(p: OracleParams, pts: torch.Tensor, want_feat: bool = True):
    """SDF trunk + heads (fields/sdf_field.py:106-123).  Returns (sdf [P,1], feat [P,256] or None).  The encoding's resolution
    (multi_res, :46-50) is read off the first layer's width: 3 + 6 multi_res input columns."""
    e = nerf_encode(pts * 3.0, (p.sdf_w[0].shape[1] - 3) // 6)
    h = e
    for l in range(8):
        if l == 4:
            h = torch.cat([h, e], dim=1) / math.sqrt(2.0)
        h = softplus100(F.linear(h, p.sdf_w[l], p.sdf_b[l]))
    sdf = F.linear(h, p.sdf_head_w, p.sdf_head_b) / 3.0
    feat = F.linear(h, p.feat_w, p.feat_b) if want_feat else None
    return sdf, feat


def sdf_gradient_autograd(p: OracleParams, pts: torch.Tensor, create_graph: bool = False) -> torch.Tensor:
    """d(sdf)/d(pts) the way the reference gets it: a full forward under enable_grad and one
    reverse sweep (fields/sdf_field.py:136-148).  ``create_graph`` keeps the result differentiable, as the
    reference's training step needs for the eikonal term and the normal-dependent colour (create_graph=True, :145)."""
    if create_graph:
        x = pts if pts.requires_grad else pts.requires_grad_(True)
    else:
        x = pts.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        y, _ = sdf_forward(p, x, want_feat=True)  # .sdf() runs the whole forward incl. out_feat (:125-126)
        (g,) = torch.autograd.grad(y, x, torch.ones_like(y), create_graph=create_graph, retain_graph=create_graph)
    return g if create_graph else g.detach()


def sdf_forward_grad_analytic(p: OracleParams, pts: torch.Tensor, want_feat: bool = True):
    """One forward that keeps sigma'(z_l) = sigmoid(100 z_l), then the reverse chain by hand.
    Mathematically identical to ``sdf_gradient_autograd``; this is the structure of the HIP kernel."""
    x3 = pts * 3.0
    m = (p.sdf_w[0].shape[1] - 3) // 6          # multi_res (6 in every shipped configuration)
    n3 = p.sdf_w[3].shape[0]                    # rows of the layer in front of the skip: d_hidden - (3 + 6 multi_res) (217)
    e = nerf_encode(x3, m)
    h = e
    dact = []
    for l in range(8):
        if l == 4:
            h = torch.cat([h, e], dim=1) / math.sqrt(2.0)
        z = F.linear(h, p.sdf_w[l], p.sdf_b[l])
        t = z * 100.0
        ez = torch.exp(t)
        dact.append(torch.where(t > 20.0, torch.ones_like(t), ez / (ez + 1.0)))  # softplus_backward
        h = softplus100(z)
    sdf = F.linear(h, p.sdf_head_w, p.sdf_head_b) / 3.0
    feat = F.linear(h, p.feat_w, p.feat_b) if want_feat else None
    g = (p.sdf_head_w / 3.0).expand(pts.shape[0], -1)
    ge_skip = None
    for l in range(7, -1, -1):
        g = (g * dact[l]) @ p.sdf_w[l]
        if l == 4:
            g = g / math.sqrt(2.0)
            ge_skip = g[:, n3:]
            g = g[:, :n3]
    ge = g + ge_skip  # gradient w.r.t. the (3 + 6 m)-d embedding (39-d)
    freqs = 2.0 ** torch.linspace(0.0, m - 1.0, m, dtype=pts.dtype, device=pts.device)
    s = (x3[..., None] * freqs)                       # [P,3,m]
    gs = ge[:, 3:3 + 3 * m].reshape(-1, 3, m)
    gc = ge[:, 3 + 3 * m:3 + 6 * m].reshape(-1, 3, m)
    dx3 = ge[:, 0:3] + ((gs * torch.cos(s) + gc * torch.cos(s + math.pi / 2.0)) * freqs).sum(-1)
    return sdf, feat, dx3 * 3.0


def color_forward(p: OracleParams, pts, normals, view, feat, pls, vis=None, cue=None) -> torch.Tensor:
    """Reflectance MLP, input order [pts, enc4(view), normals, enc4(pl), feat, enc4(vis), enc4(cue)]
    (fields/reflectance_network.py:68-96); vis / cue are absent for the pl-naive model, either one for a one-hint model (:83-86)."""
    # the reflectance net's multi_res (:41-52; 4 in every shipped configuration) from the first layer's width
    extra = (1 if vis is not None else 0) + (cue.shape[-1] if cue is not None else 0)
    mv = (p.col_w[0].shape[1] - 12 - feat.shape[-1] - extra) // (2 * (6 + extra))
    parts = [pts, nerf_encode(view, mv), normals, nerf_encode(pls, mv), feat]
    if vis is not None:
        parts.append(nerf_encode(vis, mv))
    if cue is not None:
        parts.append(nerf_encode(cue, mv))
    x = torch.cat(parts, dim=-1)
    for l in range(5):
        x = F.linear(x, p.col_w[l], p.col_b[l])
        if l < 4:
            x = torch.relu(x)
    return torch.sigmoid(x)


# --------------------------------------------------------------------------------------
# sampler
# --------------------------------------------------------------------------------------
def excl_cumprod_one_minus(alpha: torch.Tensor) -> torch.Tensor:
    """prod_{k<j}(1 - alpha_k + 1e-7)  (models/neus_hint_model.py:311-312, 429-430, 521-523)."""
    ones = torch.ones_like(alpha[:, :1])
    return torch.cumprod(torch.cat([ones, 1.0 - alpha + 1e-7], dim=-1), dim=-1)[:, :-1]


def sample_pdf_det(bins: torch.Tensor, weights: torch.Tensor, n: int) -> torch.Tensor:
    """Deterministic inverse-CDF sampling (models/neus_hint_model.py:21-65 with det=True)."""
    w = weights + 1e-5
    pdf = w / w.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], dim=-1)
    u = torch.linspace(0.0, 1.0, n).to(device=bins.device, dtype=bins.dtype).expand(bins.shape[0], n).contiguous()  # fp32 linspace as in :31
    ind = torch.searchsorted(cdf, u, right=True)
    lo = (ind - 1).clamp(min=0)
    hi = ind.clamp(max=cdf.shape[-1] - 1)
    c_lo, c_hi = torch.gather(cdf, 1, lo), torch.gather(cdf, 1, hi)
    b_lo, b_hi = torch.gather(bins, 1, lo), torch.gather(bins, 1, hi)
    den = c_hi - c_lo
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    return b_lo + (u - c_lo) / den * (b_hi - b_lo)


def up_sample(o, d, z, sdf, n_new: int, inv_s: float) -> torch.Tensor:
    """One importance step at fixed sharpness (models/neus_hint_model.py:270-315)."""
    pts = o[:, None, :] + d[:, None, :] * z[..., None]
    r = torch.linalg.norm(pts, dim=-1)
    inside = (r[:, :-1] < 1.0) | (r[:, 1:] < 1.0)
    s0, s1 = sdf[:, :-1], sdf[:, 1:]
    z0, z1 = z[:, :-1], z[:, 1:]
    mid = (s0 + s1) * 0.5
    cos = (s1 - s0) / (z1 - z0 + 1e-5)
    prev = torch.cat([torch.zeros_like(cos[:, :1]), cos[:, :-1]], dim=-1)
    cos = torch.minimum(prev, cos).clip(-1e3, 0.0) * inside
    dist = z1 - z0
    c_prev = torch.sigmoid((mid - cos * dist * 0.5) * inv_s)
    c_next = torch.sigmoid((mid + cos * dist * 0.5) * inv_s)
    alpha = (c_prev - c_next + 1e-5) / (c_prev + 1e-5)
    w = alpha * excl_cumprod_one_minus(alpha)
    return sample_pdf_det(z, w, n_new)


def merge_sorted(z, z_new, sdf=None, sdf_new=None):
    """Concatenate + sort z, carrying the SDF values along (models/neus_hint_model.py:317-331)."""
    zc, idx = torch.sort(torch.cat([z, z_new], dim=-1), dim=-1)
    if sdf is None:
        return zc, None
    return zc, torch.gather(torch.cat([sdf, sdf_new], dim=-1), 1, idx)


def hierarchical_z(p: OracleParams, o, d, z, n_steps: int = 4, n_new: int = 16, full_forward: bool = True):
    """Coarse SDF + 4 x (up_sample, merge) (models/neus_hint_model.py:696-713 and :397-412)."""
    n = o.shape[0]
    # the reference's .sdf() always pays for the feature head too (fields/sdf_field.py:125-126);
    # full_forward=False skips it (value-identical, what the HIP sampler does)
    sdf = sdf_forward(p, (o[:, None, :] + d[:, None, :] * z[..., None]).reshape(-1, 3), full_forward)[0].reshape(n, -1)
    for i in range(n_steps):
        zn = up_sample(o, d, z, sdf, n_new, 64.0 * 2 ** i)
        if i + 1 < n_steps:
            sn = sdf_forward(p, (o[:, None, :] + d[:, None, :] * zn[..., None]).reshape(-1, 3), full_forward)[0]
            z, sdf = merge_sorted(z, zn, sdf, sn.reshape(n, -1))
        else:
            z, _ = merge_sorted(z, zn)
    return z


# --------------------------------------------------------------------------------------
# alpha / hints
# --------------------------------------------------------------------------------------
def inv_s_of(p: OracleParams) -> torch.Tensor:
    """exp(10 variance) clipped to [1e-6, 1e6] (models/neus_hint_model.py:110, 337)."""
    # the reference multiplies a float32 ones([P,1]) by the 0-dim exp(10 v): the product is float32 even
    # when the module was cast to float64 - kept so the fp64 goldens match to rounding.
    return torch.exp(p.variance * 10.0).float().to(p.variance.dtype).clip(1e-6, 1e6)


def alpha_from(sdf, grad, dirs, dists, inv_s, cos_anneal: float):
    """SDF -> alpha (models/neus_hint_model.py:339-356). sdf [P,1], grad/dirs [P,3], dists [P,1]."""
    true_cos = (dirs * grad).sum(-1, keepdim=True)
    iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal) + F.relu(-true_cos) * cos_anneal)
    nxt = sdf + iter_cos * dists * 0.5
    prv = sdf - iter_cos * dists * 0.5
    c_prev, c_next = torch.sigmoid(prv * inv_s), torch.sigmoid(nxt * inv_s)
    return ((c_prev - c_next + 1e-5) / (c_prev + 1e-5)).clip(0.0, 1.0)


def _sdf_and_grad(p, pts, mode, want_feat, differentiable=False):
    if mode == "as_written":
        sdf, feat = sdf_forward(p, pts, True)               # render_core :504 / get_alpha :335
        sdf2, _ = sdf_forward(p, pts, True)                 # get_alpha's own forward (:335)
        grad = sdf_gradient_autograd(p, pts, create_graph=differentiable)  # :336 -> third forward + reverse
        return sdf2, feat, grad
    sdf, feat, grad = sdf_forward_grad_analytic(p, pts, want_feat)
    return sdf, feat, grad


def visibility(p: OracleParams, pls, hit, cos_anneal=1.0, offset=1e-2, t_rand=None, mode="minimal", differentiable=False,
               n_samples=64, n_importance=64):
    """Shadow ray light -> hit point, transmittance before the last sample
    (models/neus_hint_model.py:373-432).  ``differentiable``: renderer.shadow_hint_gradient (:379) - the final alpha evaluation
    keeps its graph w.r.t. the network (second order through d sdf/dx); the sample positions are constants w.r.t. the network
    either way (importance samples are detached, :313; the coarse ones depend on the light only)."""
    n = pls.shape[0]
    dvec = hit - pls
    L = torch.linalg.norm(dvec, dim=-1, keepdim=True)
    ds = dvec / L
    z = torch.linspace(0.0, 1.0, n_samples).to(device=pls.device, dtype=pls.dtype)[None, :] * L * (1.0 - offset)
    if t_rand is not None:  # stratified jitter in training (:388-395)
        mids = 0.5 * (z[:, 1:] + z[:, :-1])
        upper = torch.cat([mids, z[:, -1:]], -1)
        lower = torch.cat([z[:, :1], mids], -1)
        z = lower + (upper - lower) * t_rand
    if n_importance > 0:                                      # :397: four steps of n_importance // 4 (get_visibility's own default, :373)
        with torch.no_grad():
            z = hierarchical_z(p, pls, ds, z, n_steps=4, n_new=n_importance // 4, full_forward=(mode == "as_written"))
    Ts = z.shape[1]
    dists = torch.cat([z[:, 1:] - z[:, :-1], (L / float(n_samples)).expand(n, 1)], dim=-1)      # sample_dist = light_norms / n_samples (:383, :417)
    mid = z + dists * 0.5
    pts = (pls[:, None, :] + ds[:, None, :] * mid[..., None]).reshape(-1, 3)
    dirs = ds[:, None, :].expand(n, Ts, 3).reshape(-1, 3)
    if mode == "as_written":
        sdf, _ = sdf_forward(p, pts, True)
        grad = sdf_gradient_autograd(p, pts, create_graph=differentiable)
    else:
        sdf, _, grad = sdf_forward_grad_analytic(p, pts, False)
    alpha = alpha_from(sdf, grad, dirs, dists.reshape(-1, 1), inv_s_of(p), cos_anneal).reshape(n, Ts)
    return excl_cumprod_one_minus(alpha)[:, -1:]


def specular_cue(hit_normal, pls, hit, d, roughness=SPEC_ROUGHNESS) -> torch.Tensor:
    """Cook-Torrance cue for the 4 roughness values (models/neus_hint_model.py:590-615)."""
    l = F.normalize(pls - hit, dim=-1)
    v = F.normalize(-d, dim=-1)
    h = F.normalize(l + v, dim=-1)
    ndl = (hit_normal * l).sum(-1).clip(0.0, 1.0)
    ndv = (hit_normal * v).sum(-1).clip(0.0, 1.0)
    ndh = (hit_normal * h).sum(-1).clip(0.0, 1.0)
    hdv = (h * v).sum(-1).clip(0.0, 1.0)
    ndh2 = torch.pow(ndh, 2)
    out = []
    for rough in roughness:       # config.renderer.specular_roughness (models/neus_hint_model.py:161, :600)
        k = (rough + 1.0) * (rough + 1.0) / 8.0
        g = ndv / (ndv * (1.0 - k) + k) * (ndl / (ndl * (1.0 - k) + k))
        a2 = rough * rough
        ndf = a2 / (math.pi * torch.pow(ndh2 * (a2 - 1.0) + 1.0, 2))
        f = 0.04 + 0.96 * torch.pow(1.0 - hdv, 5)
        out.append(ndf * g * f / (4.0 * ndv + 1e-3))
    return torch.stack(out, dim=-1)


def nerf_forward(nerf: Dict[str, torch.Tensor], pts4, views, pls):
    """The outside NeRF (fields/nerf_density_field.py:66-89): 8 x 256 ReLU layers on enc10(pts4), the input re-attached after layer 4
    (cat[input, h], so layer 5 is 340 wide), density head, feature -> cat[feature, enc4(cat[view, light])] -> 128 ReLU -> rgb.
    ``nerf``: the module's state dict (pts_linears.N.weight ...).  Returns (density [P,1], rgb pre-sigmoid [P,3])."""
    x = nerf_encode(pts4, 10)
    v = nerf_encode(torch.cat([views, pls], dim=-1), 4)
    h = x
    for i in range(8):
        h = torch.relu(F.linear(h, nerf[f"pts_linears.{i}.weight"], nerf[f"pts_linears.{i}.bias"]))
        if i == 4:
            h = torch.cat([x, h], dim=-1)
    density = F.linear(h, nerf["alpha_linear.weight"], nerf["alpha_linear.bias"])
    feat = F.linear(h, nerf["feature_linear.weight"], nerf["feature_linear.bias"])
    h = torch.relu(F.linear(torch.cat([feat, v], dim=-1), nerf["views_linears.0.weight"], nerf["views_linears.0.bias"]))
    return density, F.linear(h, nerf["rgb_linear.weight"], nerf["rgb_linear.bias"])


def outside_z(far, n_outside: int = 32, n_samples: int = 64, t_rand=None):
    """Sample positions of the background beyond the unit sphere (models/neus_hint_model.py:677-693): inverse-depth spacing,
    stratified jitter in training, ``far / flip(u) + 1 / n_samples``.  far [N,1] -> [N, n_outside]."""
    u = torch.linspace(1e-3, 1.0 - 1.0 / (n_outside + 1.0), n_outside).to(device=far.device, dtype=far.dtype)
    if t_rand is not None:
        mids = 0.5 * (u[1:] + u[:-1])
        upper, lower = torch.cat([mids, u[-1:]]), torch.cat([u[:1], mids])
        u = lower[None, :] + (upper - lower)[None, :] * t_rand
    return far / torch.flip(u, dims=[-1]) + 1.0 / n_samples


def render_outside(nerf, o, d, pl, z, sample_dist: float):
    """``render_outside`` (models/neus_hint_model.py:434-473) at the sorted positions z [N,n]: inverted-sphere parameterisation
    (p / |p|, 1 / |p|) with |p| clipped to >= 1, alpha = 1 - exp(-softplus(density) dist).  -> (alpha [N,n], colour [N,n,3])."""
    n, m = z.shape
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full((n, 1), sample_dist, dtype=z.dtype, device=z.device)], dim=-1)
    mid = z + dists * 0.5
    pts = o[:, None, :] + d[:, None, :] * mid[..., None]
    r = torch.linalg.norm(pts, ord=2, dim=-1, keepdim=True).clip(1.0, 1e10)
    pts4 = torch.cat([pts / r, 1.0 / r], dim=-1).reshape(-1, 4)
    density, col = nerf_forward(nerf, pts4, d[:, None, :].expand(n, m, 3).reshape(-1, 3), pl[:, None, :].expand(n, m, 3).reshape(-1, 3))
    alpha = 1.0 - torch.exp(-F.softplus(density.reshape(n, m)) * dists)
    return alpha, torch.sigmoid(col).reshape(n, m, 3)


def sphere_trace(p: OracleParams, o, d, iterations: int = 2000, threshold: float = 1e-4, far: float = 100.0):
    """``NeuSHintRenderer.sphere_trace`` (models/neus_hint_model.py:359-372): from the ray origins, advance each ray by the SDF
    at its point until |sdf| < threshold or the travelled depth exceeds ``far``; -> (points [N,3], depths [N,1])."""
    pts, depths = o, torch.zeros(o.shape[0], 1, dtype=o.dtype, device=o.device)
    with torch.no_grad():
        for _ in range(iterations):
            sdf = sdf_forward(p, pts, want_feat=False)[0]
            conv = (sdf.abs() < threshold) | (depths > far)
            pts = torch.where(conv, pts, pts + sdf * d)
            depths = torch.where(conv, depths, depths + sdf)
            if bool(conv.all()):
                break
    return pts, depths


# --------------------------------------------------------------------------------------
# the renderer
# --------------------------------------------------------------------------------------
def render_forward(p: OracleParams, o, d, pl, near, far, background_rgb=None, is_training=False,
                   global_step=0, anneal_end=50_000, t_rand_primary=None, t_rand_shadow=None,
                   mode="minimal", keep_intermediates=False, differentiable=False, hints=True,
                   analytic_normal=False, depth_max_weight=False, geometry_warmup_end=0,
                   depth_sphere_tracing=False, shadow_hint=None, specular_hint=None, shadow_hint_gradient=False,
                   specular_hint_gradient=False, n_shadow_importance_clip=-1, n_importance_samples=64, outside_nerf=None,
                   t_rand_outside=None, specular_roughness=SPEC_ROUGHNESS, shadow_ray_offset=1e-2, z_override=None,
                   vis_groups_override=None, cue_override=None, n_samples=64, up_sample_steps=4, n_shadow_samples=64,
                   n_shadow_importance_samples=64, vis_override=None, net_override=None, sections_override=None) -> Dict[str, torch.Tensor]:
    """``NeuSHintRenderer.forward`` with the default nr-hints config
    (models/neus_hint_model.py:653-751 -> render_core :475-651).  ``geometry_warmup_end``: while training below that step
    both hints are fed as zeros and neither the shadow march nor the cue is evaluated (:668, :577-579, :617-619).
    ``z_override`` [N,128] / ``vis_override`` [N,1] (``vis_groups_override`` [N,clip] with the partial hint) / ``cue_override`` [N,4]: test hooks that replace the three
    NON-differentiable products of the forward (sample positions :697, partial visibility hint :553-575, specular cue :589) by
    values recorded elsewhere (the HIP path's own), so that a gradient comparison isolates the arithmetic of the differentiable
    part from where the samplers happened to place their samples.
    ``sections_override`` (mid [N,T], dists [N,T]): with ``z_override``, the section mid-points and lengths themselves instead of
    re-deriving them from z - importance samples cluster to sections of 1e-5 at the surface while z ~ 3 carries a float32 ulp of
    2.4e-7, so lengths re-derived from rounded positions would be off by per cents where it matters most.
    ``net_override`` dict(sdf [P,1], grad [P,3], feat [P,256]; any subset): a further test hook - the SDF network's outputs at the
    composite samples take these VALUES while keeping their own derivatives (x <- x + stop_gradient(x_given - x)).  With the HIP
    forward's own outputs this puts the oracle's linearisation point exactly where the HIP backward linearised: at NeuS sharpness
    ~1e3 the alpha stage amplifies float32 round-off of the SDF by three orders, so two correct float32 forwards differ by 1e-4 in
    single pixels - and in the gradients that pass through them - however the samples are placed."""
    n = o.shape[0]
    dt = o.dtype
    cos_anneal = 1.0
    # ``hints`` switches both hints; shadow_hint / specular_hint override it one by one (config.renderer.shadow_hint / specular_hint)
    shadow_hint = hints if shadow_hint is None else shadow_hint
    specular_hint = hints if specular_hint is None else specular_hint
    warmup = bool(is_training and global_step < geometry_warmup_end)   # :668
    if is_training and anneal_end > 0:
        cos_anneal = min(1.0, global_step / anneal_end)       # :669-671
    sample_dist = 2.0 / n_samples                              # :673
    z = near + (far - near) * torch.linspace(0.0, 1.0, n_samples).to(device=near.device, dtype=dt)[None, :]
    if is_training:
        z = z + (t_rand_primary - 0.5) * 2.0 / n_samples       # :681-683
    if z_override is not None:
        z = z_override.to(dt)
    elif n_importance_samples > 0:                            # :696 (n_importance_samples = 0: the coarse samples are final)
        with torch.no_grad():                                 # :696-713: up_sample_steps x (n_importance // up_sample_steps) new samples
            z = hierarchical_z(p, o, d, z, n_steps=up_sample_steps, n_new=(n_importance_samples // up_sample_steps if up_sample_steps else 0),
                               full_forward=(mode == "as_written"))
    shadow_counts = dict(n_samples=n_shadow_samples, n_importance=n_shadow_importance_samples)
    T = z.shape[1]
    bg_alpha = bg_col = None
    if outside_nerf is not None:                              # renderer.use_outside_nerf (:715-724)
        z_out = outside_z(far, 32, n_samples, t_rand_outside if is_training else None)
        z_feed, _ = torch.sort(torch.cat([z, z_out], dim=-1), dim=-1)
        bg_alpha, bg_col = render_outside(outside_nerf, o, d, pl, z_feed, sample_dist)
    # ---- render_core ----
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full((n, 1), sample_dist, dtype=dt, device=z.device)], dim=-1)
    mid = z + dists * 0.5
    if sections_override is not None:
        mid, dists = sections_override[0].to(dt), sections_override[1].to(dt)
    pts = (o[:, None, :] + d[:, None, :] * mid[..., None]).reshape(-1, 3)
    dirs = d[:, None, :].expand(n, T, 3).reshape(-1, 3)
    pls = pl[:, None, :].expand(n, T, 3).reshape(-1, 3)
    # ``differentiable`` (training): the render_core graph is kept, incl. the double-backward path through d sdf/dx
    # (mode "as_written" only); sampling, depth / hit point, shadow hint and specular cue stay outside the graph as
    # in the reference (:697, :531, :379, :589).
    sdf, feat, grad = _sdf_and_grad(p, pts, mode, True, differentiable)
    if net_override is not None:
        given = lambda x, k: x + (net_override[k].to(dt).reshape(x.shape) - x).detach() if k in net_override else x
        sdf, feat, grad = given(sdf, "sdf"), given(feat, "feat"), given(grad, "grad")
    inv_s = inv_s_of(p)
    alpha = alpha_from(sdf, grad, dirs, dists.reshape(-1, 1), inv_s, cos_anneal).reshape(n, T)
    radius = torch.linalg.norm(pts, dim=-1).reshape(n, T)
    inside = (radius < 1.0).to(dt)
    if bg_alpha is not None:                                   # :516-519: NeuS inside the unit sphere, the NeRF outside and beyond
        alpha = torch.cat([alpha * inside + bg_alpha[:, :T] * (1.0 - inside), bg_alpha[:, T:]], dim=-1)
    weights_all = alpha * excl_cumprod_one_minus(alpha)        # :521-523
    wsum = weights_all.sum(-1, keepdim=True)
    weights = weights_all[:, :T]                               # neus_weights (:524): depth, hit normal
    with torch.no_grad():
        if depth_sphere_tracing:                               # DepthComputationType.SphereTracing (:527-528)
            hit, depth = sphere_trace(p, o, d, 2000, 1e-4, 100.0)
        elif depth_max_weight:                                 # DepthComputationType.MaximalWeightPoint (:534-538)
            depth = torch.gather(mid, 1, torch.argmax(weights, dim=1, keepdim=True))
            hit = o + d * depth
        else:
            depth = (mid * weights).sum(-1, keepdim=True)      # :531-533 (no_grad)
            hit = o + d * depth
        vis_samples = None
        if shadow_hint and warmup:
            vis = torch.zeros(n, 1, dtype=dt, device=o.device)                  # :577-579 (shadow_map = zeros)
        elif shadow_hint and n_shadow_importance_clip > 0:
            # partial visibility hint (:553-575): one shadow ray per group of 128 / clip samples, aimed at z_vals[:, g * ratio]
            clip = n_shadow_importance_clip
            ratio = T // clip
            zt = z[:, torch.arange(0, T, ratio, device=z.device)]
            tgt = (o[:, None, :] + d[:, None, :] * zt[..., None]).reshape(-1, 3)
            pls_g = pl[:, None, :].repeat(1, clip, 1).reshape(-1, 3)
            if vis_groups_override is not None:
                vg = vis_groups_override.to(dt).reshape(n, clip, 1)
            else:
                vg = visibility(p, pls_g, tgt, cos_anneal, shadow_ray_offset, t_rand_shadow if is_training else None, mode,
                                **shadow_counts).reshape(n, clip, 1)
            vis_samples = vg.repeat_interleave(ratio, dim=1)                               # [n,128,1]
            vis = torch.gather(vis_samples[..., 0], 1, torch.argmax(weights, dim=1, keepdim=True))   # shadow_map (:573-574)
        elif shadow_hint and vis_override is not None:
            vis = vis_override.to(dt).reshape(n, 1)
        elif not (shadow_hint and shadow_hint_gradient and differentiable):
            vis = visibility(p, pl, hit, cos_anneal, shadow_ray_offset, t_rand_shadow if is_training else None, mode, **shadow_counts) \
                if shadow_hint else None                       # :546-551, :379
    if shadow_hint and not warmup and shadow_hint_gradient and differentiable:
        vis = visibility(p, pl, hit, cos_anneal, shadow_ray_offset, t_rand_shadow if is_training else None, mode, differentiable=True,
                         **shadow_counts)
    n_hat = F.normalize(grad, dim=-1)                          # :584
    hit_n = F.normalize((n_hat.reshape(n, T, 3) * weights[..., None]).sum(1), dim=-1)  # :586-587
    vis_s = cue_s = None
    if shadow_hint:
        vis_s = vis[:, None, :].expand(n, T, 1).reshape(-1, 1) if vis_samples is None else vis_samples.reshape(-1, 1)
    if specular_hint:
        with torch.enable_grad() if (specular_hint_gradient and differentiable) else torch.no_grad():   # :589
            cue = torch.zeros(n, 4, dtype=dt, device=o.device) if warmup else specular_cue(hit_n, pl, hit, d, specular_roughness)   # :590-615, :617-619
            if cue_override is not None:
                cue = cue_override.to(dt)
        cue_s = cue[:, None, :].expand(n, T, 4).reshape(-1, 4)
    col = color_forward(p, pts, grad if analytic_normal else n_hat, dirs, feat, pls, vis_s, cue_s).reshape(n, T, 3)  # :621-626
    if bg_col is not None:                                     # :630-633
        col_all = torch.cat([col * inside[..., None] + bg_col[:, :T] * (1.0 - inside)[..., None], bg_col[:, T:]], dim=1)
    else:
        col_all = col
    rgb = (col_all * weights_all[..., None]).sum(1)
    if background_rgb is not None:
        rgb = rgb + background_rgb * (1.0 - wsum)              # :635-637
    out = dict(rgb=rgb, depth=depth, weights=weights_all, s_val=(1.0 / inv_s).expand(n, T),
               inside_sphere=inside, relax_inside_sphere=inside,            # :745 (quirk kept)
               analytic_normals=grad.reshape(n, T, 3),
               normalized_analytic_normals=n_hat.reshape(n, T, 3),
               visibilities=vis, specular_cue=cue_s.reshape(n, T, 4) if specular_hint else None)
    if keep_intermediates:
        out.update(z_vals=z, mid_z=mid, sdf=sdf.reshape(n, T), alpha=alpha, hit=hit, hit_normal=hit_n,
                   sampled_color=col, feat=feat)
    return out


def render_chunked(p, o, d, pl, near, far, chunk=512, **kw):
    """The reference's eval loop shape: 512-ray chunks (models/neus_hint_model.py:212,
    pipelines/base_pipeline.py:110-120)."""
    outs = [render_forward(p, o[i:i + chunk], d[i:i + chunk], pl[i:i + chunk], near[i:i + chunk],
                           far[i:i + chunk], **kw) for i in range(0, o.shape[0], chunk)]
    return {k: (torch.cat([x[k] for x in outs], dim=0) if outs[0][k] is not None else None) for k in outs[0]}


def train_loss(out, rgb_gt, igr_weight=0.1):
    """L1 colour + eikonal (pipelines/base_pipeline.py:57-62)."""
    nrays = out["rgb"].shape[0]
    rgb_loss = (out["rgb"] - rgb_gt).abs().sum() / (nrays + 1e-5)
    ge = (torch.linalg.norm(out["analytic_normals"], dim=-1) - 1.0) ** 2
    m = out["relax_inside_sphere"]
    eik = (m * ge).sum() / (m.sum() + 1e-5)
    return rgb_loss + igr_weight * eik, rgb_loss, eik
