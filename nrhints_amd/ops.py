"""Thin tensor-level wrappers over the fine-grained C entry points (include/nrhints_hip.h).

``NeuSHintRenderer.forward`` does not use these (it makes one fused C call per chunk); they exist so that each
kernel can be parity-tested in isolation and so that callers such as mesh extraction can query the SDF alone.
All tensors are float32, contiguous, on the GPU.  No fallback: errors raise.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib, packing


def _scratch(device) -> torch.Tensor:
    lib = _lib.load()
    n = lib.nrh_mlp_grid() * _lib.param_sizes()[7] * packing.SDF_SCRATCH_FLOATS_PER_WAVE
    return torch.empty(n, dtype=torch.float32, device=device)


def _wptr(w):
    """Packed weights: float32 (precision 0) or float16 pairs (precision 1, f16x3)."""
    return _lib.ptr(w, w.dtype), (0 if w.dtype == torch.float32 else 1)


@_lib.on_tensor_device
def sdf_eval(mode: int, sdf_w, sdf_b, sdf_head, ro, rd, t, n_per_ray: int, t_stride: Optional[int] = None,
             scratch: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor]]:
    """SDF (mode 0), + gradient (mode 1), + feature (mode 2) at points ro[ray] + rd[ray] * t[ray, j].

    Returns (sdf [nrays, n_per_ray], grad [nrays*n_per_ray, 3] | None, feat tiles | None)."""
    lib = _lib.load()
    nrays = ro.shape[0]
    t_stride = n_per_ray if t_stride is None else t_stride
    dev = ro.device
    sdf = torch.empty(nrays, n_per_ray, dtype=torch.float32, device=dev)
    npts = nrays * n_per_ray
    grad = torch.empty(npts, 3, dtype=torch.float32, device=dev) if mode >= 1 else None
    feat = torch.empty(((npts + 15) // 16) * 4096, dtype=torch.float32, device=dev) if mode == 2 else None
    if mode >= 1 and scratch is None:
        scratch = _scratch(dev)
    P = _lib.ptr
    wp, prec = _wptr(sdf_w)
    rc = lib.nrh_sdf_eval(prec, mode, wp, P(sdf_b), P(sdf_head), P(ro), P(rd), P(t), t_stride, n_per_ray, nrays,
                          P(sdf), n_per_ray, P(grad), P(feat), P(scratch) if mode >= 1 else None, _lib.stream_handle())
    _lib.check(rc, "nrh_sdf_eval")
    return sdf, grad, feat


@_lib.on_tensor_device
def sdf_eval_wide(mode: int, sdf_w32, sdf_tab32, ro, rd, t, n_per_ray: int, t_stride: Optional[int] = None,
                  scratch: Optional[torch.Tensor] = None, one_term: bool = False) -> Tuple[torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor]]:
    """``sdf_eval`` through the wide f16x3 kernels (csrc/nrh_sdf32.hip); sdf_w32 / sdf_tab32 from packing32.pack_sdf32.
    mode 3 (wide only): sdf + the derivative along the ray in forward mode; ``grad`` = rd * (d sdf / dt) / |rd|^2.
    ``one_term``: their single-pass builds (precision "f16": one fp16 MFMA per K step, csrc/nrh_wide1.hip), same buffers."""
    lib = _lib.load()
    nrays = ro.shape[0]
    t_stride = n_per_ray if t_stride is None else t_stride
    dev = ro.device
    sdf = torch.empty(nrays, n_per_ray, dtype=torch.float32, device=dev)
    npts = nrays * n_per_ray
    grad = torch.empty(npts, 3, dtype=torch.float32, device=dev) if mode >= 1 else None
    feat = torch.empty(((npts + 15) // 16) * 4096, dtype=torch.float32, device=dev) if mode == 2 else None
    if mode in (1, 2) and scratch is None:
        scratch = _scratch(dev)
    if sdf_w32.numel() * sdf_w32.element_size() != lib.nrh_sdf_wide_stream_bytes():
        raise ValueError("sdf_w32 has the wrong size for this library build")
    P = _lib.ptr
    with torch.cuda.device(dev):
        fn = lib.nrh_sdf_eval_wide_f16 if one_term else lib.nrh_sdf_eval_wide
        rc = fn(mode, P(sdf_w32, sdf_w32.dtype), P(sdf_tab32), P(ro), P(rd), P(t), t_stride, n_per_ray, nrays,
                P(sdf), n_per_ray, P(grad), P(feat), P(scratch) if mode in (1, 2) else None, _lib.stream_handle(dev))
    _lib.check(rc, "nrh_sdf_eval_wide")
    return sdf, grad, feat


@_lib.on_tensor_device
def sdf_eval_split(sdf_w, sdf_b, sdf_head, ro, rd, t, n_per_ray: int, t_stride: Optional[int] = None, tiles: int = 0) -> torch.Tensor:
    """SDF values of a small point set on the channel-split kernel (csrc/nrh_sdf_split.hip): ``sdf_eval`` mode 0 with the
    precision-1 (f16x3) packed parameters, bit-identical values, a quarter of the per-pass latency.  tiles: 0 (auto), 1 or 2."""
    lib = _lib.load()
    nrays = ro.shape[0]
    t_stride = n_per_ray if t_stride is None else t_stride
    sdf = torch.empty(nrays, n_per_ray, dtype=torch.float32, device=ro.device)
    P = _lib.ptr
    with torch.cuda.device(ro.device):
        rc = lib.nrh_sdf_eval_split(P(sdf_w, sdf_w.dtype), P(sdf_b), P(sdf_head), P(ro), P(rd), P(t), t_stride, n_per_ray, nrays,
                                    P(sdf), n_per_ray, tiles, _lib.stream_handle(ro.device))
    _lib.check(rc, "nrh_sdf_eval_split")
    return sdf


@_lib.on_tensor_device
def sdf_grad_split(sdf_w, sdf_b, sdf_head, ro, rd, t, n_per_ray: int, t_stride: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """sdf and d sdf / dx of a small point set on the channel-split kernel: ``sdf_eval`` mode 1 with the precision-1 (f16x3) packed
    parameters, bit-identical values, no scratch.  -> (sdf [nrays, n_per_ray], grad [nrays * n_per_ray, 3])."""
    lib = _lib.load()
    nrays = ro.shape[0]
    t_stride = n_per_ray if t_stride is None else t_stride
    sdf = torch.empty(nrays, n_per_ray, dtype=torch.float32, device=ro.device)
    grad = torch.empty(nrays * n_per_ray, 3, dtype=torch.float32, device=ro.device)
    P = _lib.ptr
    with torch.cuda.device(ro.device):
        rc = lib.nrh_sdf_grad_split(P(sdf_w, sdf_w.dtype), P(sdf_b), P(sdf_head), P(ro), P(rd), P(t), t_stride, n_per_ray, nrays,
                                    P(sdf), n_per_ray, P(grad), _lib.stream_handle(ro.device))
    _lib.check(rc, "nrh_sdf_grad_split")
    return sdf, grad


@_lib.on_tensor_device
def sdf_at_points(mode: int, sdf_w, sdf_b, sdf_head, pts):
    """Convenience: free points [P,3] (one 'ray' per point, t = 0)."""
    zeros = torch.zeros_like(pts)
    t = torch.zeros(pts.shape[0], dtype=torch.float32, device=pts.device)
    return sdf_eval(mode, sdf_w, sdf_b, sdf_head, pts.contiguous(), zeros, t, 1)


@_lib.on_tensor_device
def sdf_train_forward(sdf_w, sdf_b, sdf_head, pts, half_handoffs: bool = False):
    """Training forward at free points [P,3] (P % 16 == 0): -> (sdf [P,1], feat [P,256] row-major, grad [P,3], saves)
    where ``saves`` holds what ``sdf_train_backward`` and the weight-gradient GEMMs need (include/nrhints_hip.h).
    ``half_handoffs`` (f16x3, nrh_train_half_supported): saves also has ``h16`` (layers 0..6; saves["h"] then only holds layer 7) and
    ``t16`` (layers 1..7), float16 [8,P,256] in the half-tiled layout (nrh_sdf_train_forward_half)."""
    lib = _lib.load()
    n = pts.shape[0]
    dev = pts.device
    f32 = dict(dtype=torch.float32, device=dev)
    zeros3 = torch.zeros(n, 3, **f32)
    zeros1 = torch.zeros(n, **f32)
    sdf = torch.empty(n, 1, **f32)
    grad = torch.empty(n, 3, **f32)
    feat = torch.empty(n, 256, **f32)
    saves = dict(h=torch.empty(8, n, 256, **f32), s1=torch.empty(8, n, 256, **f32), t=torch.empty(8, n, 256, **f32),
                 ge=torch.empty(n, 128, **f32), zeros3=zeros3, zeros1=zeros1)
    P = _lib.ptr
    wp, prec = _wptr(sdf_w)
    if half_handoffs:
        saves["h16"] = torch.empty(8, n, 256, dtype=torch.float16, device=dev)
        saves["t16"] = torch.empty(8, n, 256, dtype=torch.float16, device=dev)
        rc = lib.nrh_sdf_train_forward_half(prec, wp, P(sdf_b), P(sdf_head), P(pts), P(zeros3), P(zeros1), 1, 1, n, P(sdf), P(grad),
                                            P(feat), P(saves["h"]), P(saves["s1"]), P(saves["t"]), P(saves["ge"]),
                                            P(saves["h16"], torch.float16), P(saves["t16"], torch.float16), _lib.stream_handle())
        _lib.check(rc, "nrh_sdf_train_forward_half")
        return sdf, feat, grad, saves
    rc = lib.nrh_sdf_train_forward(prec, wp, P(sdf_b), P(sdf_head), P(pts), P(zeros3), P(zeros1), 1, 1, n, P(sdf), P(grad),
                                   P(feat), P(saves["h"]), P(saves["s1"]), P(saves["t"]), P(saves["ge"]), _lib.stream_handle())
    _lib.check(rc, "nrh_sdf_train_forward")
    return sdf, feat, grad, saves


@_lib.on_tensor_device
def sdf_train_backward(sdf_w, wt_feat, sdf_head, ro, rd, t, n_per_ray, saves, sbar, fbar, gbar, adj_scale: float = 1.0,
                       half_handoffs: bool = False, dyn: Optional[torch.Tensor] = None):
    """The two backward sweeps at the points ro[ray] + rd[ray] * t[ray, j]
    -> dict(abar, coup, zbar [8,P,256], gebar [P,64], pbar [P,3]).  ``adj_scale``: see _lib.adjoint_scale (a training step passes
    it; 1 = the adjoints as they come, for callers whose adjoints are O(1)).
    ``half_handoffs`` (f16x3, batches with nrh_train_half_supported): layers 0..6 of abar and 1..7 of zbar leave as float16 arrays
    ``abar16`` / ``zbar16`` (half-tiled, times the step's adjoint scale, which the library takes from the seeds' range and writes to
    ``dyn`` = float32 [4], zero before its first use) INSTEAD of the float32 arrays; adj_scale is not used."""
    lib = _lib.load()
    nrays = ro.shape[0]
    n = nrays * n_per_ray
    f32 = dict(dtype=torch.float32, device=ro.device)
    out = dict(abar=torch.empty(8, n, 256, **f32), coup=torch.empty(8, n, 256, **f32), zbar=torch.empty(8, n, 256, **f32),
               gebar=torch.empty(n, 64, **f32), pbar=torch.empty(n, 3, **f32))
    P = _lib.ptr
    wp, prec = _wptr(sdf_w)
    wtp, prec2 = _wptr(wt_feat)
    if prec != prec2:
        raise ValueError("sdf_w and wt_feat are packed for different precisions")
    if half_handoffs:
        out["abar16"] = torch.empty(8, n, 256, dtype=torch.float16, device=ro.device)
        out["zbar16"] = torch.empty(8, n, 256, dtype=torch.float16, device=ro.device)
        out["dyn"] = dyn if dyn is not None else torch.zeros(4, **f32)
        rc = lib.nrh_sdf_train_backward_half(prec, wp, wtp, P(sdf_head), P(ro), P(rd), P(t), n_per_ray, n_per_ray, nrays,
                                             P(saves["s1"]), P(saves["t"]), P(gbar), P(fbar), P(sbar), P(out["abar"]), P(out["coup"]),
                                             P(out["gebar"]), P(out["zbar"]), P(out["pbar"]), P(out["abar16"], torch.float16),
                                             P(out["zbar16"], torch.float16), P(out["dyn"]), P(saves.get("t16"), torch.float16),
                                             _lib.stream_handle())
        _lib.check(rc, "nrh_sdf_train_backward_half")
        return out
    rc = lib.nrh_sdf_train_backward(prec, wp, wtp, P(sdf_head), P(ro), P(rd), P(t), n_per_ray, n_per_ray, nrays,
                                    P(saves["s1"]), P(saves["t"]), P(gbar), P(fbar), P(sbar), P(out["abar"]), P(out["coup"]),
                                    P(out["gebar"]), P(out["zbar"]), P(out["pbar"]), float(adj_scale), _lib.stream_handle())
    _lib.check(rc, "nrh_sdf_train_backward")
    return out


@_lib.on_tensor_device
def sampler_step(ro, rd, z, s, n: int, *, znew_in=None, snew_in=None, upsample_inv_s: Optional[float] = None,
                 lin16=None, finalize: bool = False, last_dist: float = 2.0 / 64, last_dist_ray=None):
    """One launch of the hierarchical sampler.  ``z``/``s`` ([nrays,128]) are updated in place by a merge.
    Returns (znew_out [nrays,16] | None, tmid | None, dists | None)."""
    lib = _lib.load()
    nrays = ro.shape[0]
    dev = ro.device
    do_merge = znew_in is not None
    merge_sdf = snew_in is not None
    do_up = upsample_inv_s is not None
    znew_out = torch.empty(nrays, 16, dtype=torch.float32, device=dev) if do_up else None
    tmid = torch.empty(nrays, 128, dtype=torch.float32, device=dev) if finalize else None
    dists = torch.empty(nrays, 128, dtype=torch.float32, device=dev) if finalize else None
    P = _lib.ptr
    rc = lib.nrh_sampler_step(P(ro), P(rd), P(z), P(s), P(znew_in), P(snew_in), P(znew_out), P(lin16),
                              P(last_dist_ray), P(tmid), P(dists), float(upsample_inv_s or 0.0), float(last_dist),
                              nrays, n, int(do_merge), int(merge_sdf), int(do_up), int(finalize), _lib.stream_handle())
    _lib.check(rc, "nrh_sampler_step")
    return znew_out, tmid, dists


@_lib.on_tensor_device
def color_eval(col_w, col_b, feat_tiles, ro, rd, tmid, nhat, raymisc, hints: bool = True) -> torch.Tensor:
    """Reflectance MLP for nrays x 128 samples -> [nrays*128, 3] (``hints=False``: the 316-input pl-naive net)."""
    lib = _lib.load()
    nrays = ro.shape[0]
    color = torch.empty(nrays * 128, 3, dtype=torch.float32, device=ro.device)
    P = _lib.ptr
    wp, prec = _wptr(col_w)
    rc = lib.nrh_color_eval(prec, int(hints), wp, P(col_b), P(feat_tiles), P(ro), P(rd), P(tmid), P(nhat), P(raymisc), nrays,
                            P(color), _lib.stream_handle())
    _lib.check(rc, "nrh_color_eval")
    return color


@_lib.on_tensor_device
def color_eval_wide(col_w32, col_tab32, part_tiles, ro, rd, tmid, nhat, raymisc) -> torch.Tensor:
    """Reflectance MLP on the wide kernel for nrays x 128 samples -> [nrays*128, 3]; ``part_tiles``: W0feat * feature as
    16-point D-layout tiles (packing.rows_to_feat_tiles); ``raymisc`` [nrays, 100]."""
    lib = _lib.load()
    nrays = ro.shape[0]
    color = torch.empty(nrays * 128, 3, dtype=torch.float32, device=ro.device)
    P = _lib.ptr
    rc = lib.nrh_color_eval_wide(P(col_w32, col_w32.dtype), P(col_tab32), P(part_tiles), P(ro), P(rd), P(tmid), P(nhat), P(raymisc),
                                 nrays, P(color), _lib.stream_handle())
    _lib.check(rc, "nrh_color_eval_wide")
    return color


@_lib.on_tensor_device
def alpha_composite(ro, rd, pl, sdf, grad, dists, mid_z, inv_s: float, cos_anneal: float = 1.0, depth_type: int = 0,
                    zero_hints: bool = False, t_rand_shadow=None):
    """The evaluation render's alpha / compositing / hit point / specular-cue stage on its own (nrh_alpha_composite =
    core_alpha_kernel) -> dict(weights, inside [n,128], nhat [n*128,3], depth, wsum [n], cue [n,4], hit, hit_normal [n,3],
    shadow_dirs [n,3], shadow_last_dist [n], shadow_z [n,128])."""
    lib = _lib.load()
    n = ro.shape[0]
    new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=ro.device)
    lin64 = torch.linspace(0.0, 1.0, 64).to(ro.device)
    out = dict(weights=new(n, 128), inside=new(n, 128), nhat=new(n * 128, 3), depth=new(n), wsum=new(n), cue=new(n, 4),
               hit=new(n, 3), hit_normal=new(n, 3), shadow_dirs=new(n, 3), shadow_last_dist=new(n), shadow_z=torch.zeros(n, 128, dtype=torch.float32, device=ro.device))
    P = _lib.ptr
    rc = lib.nrh_alpha_composite(P(ro), P(rd), P(pl), P(sdf), P(grad), P(dists), P(mid_z), float(inv_s), float(cos_anneal), int(depth_type),
                                 int(bool(zero_hints)), P(lin64), P(t_rand_shadow), n, P(out["weights"]), P(out["inside"]), P(out["nhat"]),
                                 P(out["depth"]), P(out["wsum"]), P(out["cue"]), P(out["hit"]), P(out["hit_normal"]), P(out["shadow_dirs"]),
                                 P(out["shadow_last_dist"]), P(out["shadow_z"]), _lib.stream_handle())
    _lib.check(rc, "nrh_alpha_composite")
    return out


@_lib.on_tensor_device
def visibility(rd, pl, shadow_dirs, sdf, grad, dists, cue, inv_s: float, cos_anneal: float = 1.0, zero_hints: bool = False):
    """Shadow-ray transmittance + the reflectance net's per-ray encodings (nrh_visibility = shadow_finish_kernel)
    -> (visibilities [n], raymisc [n, RAYMISC_STRIDE])."""
    from . import packing
    lib = _lib.load()
    n = rd.shape[0]
    vis = torch.empty(n, dtype=torch.float32, device=rd.device)
    raymisc = torch.zeros(n, packing.RAYMISC_STRIDE, dtype=torch.float32, device=rd.device)
    P = _lib.ptr
    rc = lib.nrh_visibility(P(rd), P(pl), P(shadow_dirs), P(sdf), P(grad), P(dists), P(cue), float(inv_s), float(cos_anneal),
                            int(bool(zero_hints)), n, P(vis), P(raymisc), _lib.stream_handle())
    _lib.check(rc, "nrh_visibility")
    return vis, raymisc


@_lib.on_tensor_device
def color_composite(color, weights, wsum, background=None, inside=None, grad=None, nhat=None, maps: bool = False):
    """rgb = sum c w + bg (1 - sum w) (+ the two weighted normal maps) (nrh_color_composite = composite_kernel)
    -> (rgb [n,3], normal_map | None, normalized_normal_map | None)."""
    lib = _lib.load()
    n = weights.shape[0]
    new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=weights.device)
    rgb = new(n, 3)
    nm, nnm = (new(n, 3), new(n, 3)) if maps else (None, None)
    P = _lib.ptr
    rc = lib.nrh_color_composite(P(color), P(weights), P(wsum), P(background), P(inside), P(grad), P(nhat), n, P(rgb), P(nm), P(nnm),
                                 _lib.stream_handle())
    _lib.check(rc, "nrh_color_composite")
    return rgb, nm, nnm
