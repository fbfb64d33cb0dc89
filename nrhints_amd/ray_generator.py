"""Pixel -> ray generation with learnable per-view pose / light refinement.

Counterpart of ``RayGenerator`` (camera/ray_generator.py:42-150) and of the two exponential maps it uses
(camera/lie_groups.py:26-122): same config fields, same parameter / buffer names (``cam_pose_adjustment`` [V,6] as
(translation, so(3) vector), ``pl_adjustment`` [V,3], ``cam_pose_noise``, ``pl_noise``) so optimiser groups and
checkpoints carry over (pipelines/base_pipeline.py:35-39, 103-105).

On the GPU a bundle is one HIP launch (``nrh_generate_rays_indexed``): the per-VIEW deltas - noise, then the exp map of
the adjustment, [V,3,4] - are composed with a handful of tiny torch ops (differentiable, V ~ 100), the per-RAY work
(pose composition, pixel -> direction, normalisation, light offset, near/far) runs in the kernel, and its adjoint kernel
scatters the ray gradients the HIP training path provides (origins / directions / pl_positions / nears / fars) back into
d/d(delta) with atomics.  CPU tensors take the same formulas as torch ops (the form the CPU parity test checks against the
reference's fixture).  Whole evaluation views without refinement use ``pipeline.generate_rays``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Literal

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from .containers import RawPixelBundle, RayBundle
from .pipeline import CameraModel


@dataclass(frozen=True)
class RayGeneratorConfig:
    """Field for field camera/ray_generator.py:14-38."""
    override_near_far_from_sphere: bool = True
    cam_opt_mode: Literal["off", "SO3xR3", "SE3"] = "off"
    pl_opt: bool = False
    opt_lr: float = 3e-5
    cam_position_noise_std: float = 0.0
    cam_orientation_noise_std: float = 0.0
    pl_position_noise_std: float = 0.0


def _hat(w: torch.Tensor) -> torch.Tensor:
    """so(3) vectors [B,3] -> skew-symmetric matrices [B,3,3] with hat(w) x = w cross x."""
    z = torch.zeros_like(w[:, 0])
    return torch.stack([z, -w[:, 2], w[:, 1], w[:, 2], z, -w[:, 0], -w[:, 1], w[:, 0], z], dim=-1).reshape(-1, 3, 3)


def _mm3(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Batched [B,3,3] x [B,3,k] as a broadcast product + sum over the inner index: a handful of 3 x 3 products per view do not
    warrant a library GEMM call (and keep rocBLAS out of the training process)."""
    return (a[:, :, :, None] * b[:, None, :, :]).sum(2)


def exp_map_SO3xR3(tangent: torch.Tensor) -> torch.Tensor:
    """[B,6] (translation, rotation vector) -> [B,3,4] = [R | t] with R = I + sin(th)/th K + (1-cos th)/th^2 K^2,
    th = sqrt(max(|w|^2, 1e-4)) (the reference clamps the SQUARED norm, camera/lie_groups.py:39-43), t copied."""
    w = tangent[:, 3:]
    K = _hat(w)
    th = (w * w).sum(1).clamp(min=1e-4).sqrt()
    a = (th.sin() / th)[:, None, None]
    b = ((1.0 - th.cos()) / (th * th))[:, None, None]
    eye = torch.eye(3, dtype=w.dtype, device=w.device)[None]
    R = eye + a * K + b * _mm3(K, K)
    return torch.cat([R, tangent[:, :3, None]], dim=-1)


def exp_map_SE3(tangent: torch.Tensor) -> torch.Tensor:
    """[B,6] -> [B,3,4], the se(3) exponential with the reference's small-angle branches (theta < 1e-2,
    camera/lie_groups.py:80-122): rational cosine, half-angle series for the other coefficients."""
    v, w = tangent[:, :3], tangent[:, 3:]
    th = torch.linalg.norm(w, dim=1, keepdim=True)          # [B,1]
    th2 = th * th
    small = th < 1e-2
    one = torch.ones_like(th)
    th_s, th2_s, th3_s = torch.where(small, one, th), torch.where(small, one, th2), torch.where(small, one, th2 * th)
    sin = th.sin()
    cos = torch.where(small, 8.0 / (4.0 + th2) - 1.0, th.cos())
    a = torch.where(small, 0.5 * cos + 0.5, sin / th_s)                 # sin(th)/th
    b = torch.where(small, 0.5 * a, (1.0 - cos) / th2_s)                # (1-cos th)/th^2
    eye = torch.eye(3, dtype=w.dtype, device=w.device)[None]
    R = b[:, :, None] * (w[:, :, None] * w[:, None, :]) + cos[:, :, None] * eye + a[:, :, None] * _hat(w)
    a_t = torch.where(small, 1.0 - th2 / 6.0, a)
    b_t = torch.where(small, 0.5 - th2 / 24.0, b)
    c_t = torch.where(small, 1.0 / 6.0 - th2 / 120.0, (th - sin) / th3_s)
    t = a_t * v + b_t * torch.linalg.cross(w, v, dim=1) + c_t * w * (w * v).sum(1, keepdim=True)
    return torch.cat([R, t[:, :, None]], dim=-1)


class _RaysIndexedHip(torch.autograd.Function):
    """nrh_generate_rays_indexed / _backward (csrc/nrh_rays.hip: raygen_indexed_kernel + adjoint)."""

    @staticmethod
    def forward(ctx, delta, pl_delta, idx, h, w, poses, pls, geom):
        lib, P = _lib.load(), _lib.ptr
        n, dev = h.shape[0], h.device
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        o, d, p, near, far = new(n, 3), new(n, 3), new(n, 3), new(n, 1), new(n, 1)
        ncam = 0 if delta is None and pl_delta is None else int((delta if delta is not None else pl_delta).shape[0])
        cx, cy, fx, fy, sphere, zn, zf = geom
        ctx.geom, ctx.ncam = geom, ncam
        ctx.save_for_backward(delta, pl_delta, idx, h, w, poses, pls)
        with torch.cuda.device(dev):
            rc = lib.nrh_generate_rays_indexed(P(idx, torch.int64), P(h), P(w), P(poses), poses.shape[-1] * poses.shape[-2], P(pls), n,
                                               P(delta), P(pl_delta), ncam, cx, cy, fx, fy, int(sphere), zn, zf,
                                               P(o), P(d), P(p), P(near), P(far), _lib.stream_handle())
        _lib.check(rc, "nrh_generate_rays_indexed")
        return o, d, p, near, far

    @staticmethod
    def backward(ctx, go, gd, gp, gn, gf):
        delta, pl_delta, idx, h, w, poses, pls = ctx.saved_tensors
        lib, P = _lib.load(), _lib.ptr
        cx, cy, fx, fy, sphere, zn, zf = ctx.geom
        c = lambda g: None if g is None else g.contiguous().float()
        go, gd, gp, gn, gf = c(go), c(gd), c(gp), c(gn), c(gf)
        g_delta = torch.zeros_like(delta) if (delta is not None and ctx.needs_input_grad[0]) else None
        g_pl = torch.zeros_like(pl_delta) if (pl_delta is not None and ctx.needs_input_grad[1]) else None
        if g_delta is not None or g_pl is not None:
            with torch.cuda.device(h.device):
                rc = lib.nrh_generate_rays_indexed_backward(P(idx, torch.int64), P(h), P(w), P(poses), poses.shape[-1] * poses.shape[-2], P(pls),
                                                            h.shape[0], P(delta), P(pl_delta), ctx.ncam, cx, cy, fx, fy, int(sphere),
                                                            P(go), P(gd), P(gp), P(gn), P(gf), P(g_delta), P(g_pl),
                                                            _lib.stream_handle())
            _lib.check(rc, "nrh_generate_rays_indexed_backward")
        return g_delta, g_pl, None, None, None, None, None, None


class RayGenerator(nn.Module):
    """``forward(RawPixelBundle) -> RayBundle`` (camera/ray_generator.py:75-150)."""

    def __init__(self, camera: CameraModel, num_cameras: int, config: RayGeneratorConfig = None, zn: float = 0.1,
                 zf: float = 10.0):
        super().__init__()
        self.camera = camera
        self.config = cfg = RayGeneratorConfig() if config is None else config
        self.zn, self.zf = float(getattr(camera, "zn", zn)), float(getattr(camera, "zf", zf))
        if cfg.cam_opt_mode not in ("off", "SO3xR3", "SE3"):
            raise ValueError(f"Unknown camera pose optimization mode: {cfg.cam_opt_mode}")
        if cfg.cam_opt_mode != "off":
            self.cam_pose_adjustment = nn.Parameter(torch.zeros(num_cameras, 6))
        if cfg.pl_opt:
            self.pl_adjustment = nn.Parameter(torch.zeros(num_cameras, 3))
        # synthetic-experiment noise, drawn once in the reference's order (pose first, then light; :61-73)
        if cfg.cam_position_noise_std != 0.0 or cfg.cam_orientation_noise_std != 0.0:
            if cfg.cam_position_noise_std < 0.0 or cfg.cam_orientation_noise_std < 0.0:
                raise ValueError("noise stds must be >= 0")
            std = torch.tensor([[cfg.cam_position_noise_std] * 3 + [cfg.cam_orientation_noise_std] * 3], dtype=torch.float32)
            self.register_buffer("cam_pose_noise", exp_map_SE3(torch.normal(torch.zeros(num_cameras, 6), std)), persistent=True)
        if cfg.pl_position_noise_std != 0.0:
            if cfg.pl_position_noise_std < 0.0:
                raise ValueError("noise stds must be >= 0")
            self.register_buffer("pl_noise", torch.normal(torch.zeros(num_cameras, 3), cfg.pl_position_noise_std), persistent=True)

    @staticmethod
    def _compose(delta: torch.Tensor, R: torch.Tensor, t: torch.Tensor):
        """Left-multiply the camera-to-world [R|t] by a [B,3,4] delta."""
        dR, dt = delta[:, :3, :3], delta[:, :3, 3:]
        return _mm3(dR, R), dt + _mm3(dR, t)

    def view_deltas(self):
        """Per-view left deltas, composed in the reference's order - noise first, then the adjustment (:108-121), i.e.
        delta = exp(adjustment) o noise - as ([V,3,4] or None, [V,3] or None)."""
        cfg, delta, pl_delta = self.config, None, None
        if hasattr(self, "cam_pose_noise"):
            delta = self.cam_pose_noise
        if cfg.cam_opt_mode != "off":
            adj = (exp_map_SO3xR3 if cfg.cam_opt_mode == "SO3xR3" else exp_map_SE3)(self.cam_pose_adjustment)
            delta = adj if delta is None else torch.cat(self._compose(adj, delta[:, :3, :3], delta[:, :3, 3:]), dim=-1)
        if hasattr(self, "pl_noise"):
            pl_delta = self.pl_noise
        if cfg.pl_opt:
            pl_delta = self.pl_adjustment if pl_delta is None else pl_delta + self.pl_adjustment
        return delta, pl_delta

    def num_views(self) -> int:
        """rows of the per-view delta tables (0: the generator refines nothing)"""
        for name in ("cam_pose_adjustment", "pl_adjustment", "cam_pose_noise", "pl_noise"):
            if hasattr(self, name):
                return int(getattr(self, name).shape[0])
        return 0

    def validate_view_indices(self, img_indices: torch.Tensor) -> None:
        """Raise ``IndexError`` - as the reference's table lookup does - if a view index lies outside the delta tables.  One host
        sync: call it once per dataset (e.g. on the data loader's index tensor), not per batch."""
        nv = self.num_views()
        if nv == 0 or img_indices is None or img_indices.numel() == 0:
            return
        lo, hi = (int(v) for v in torch.stack([img_indices.min(), img_indices.max()]).tolist())
        if lo < -nv or hi >= nv:
            raise IndexError(f"view index out of range: the ray generator holds {nv} views, the bundle has indices in [{lo}, {hi}]")
        if lo < 0:
            raise IndexError(f"negative view index {lo}: the HIP ray generator takes indices in [0, {nv})")

    def _forward_hip(self, pb: RawPixelBundle) -> RayBundle:
        cam, cfg = self.camera, self.config
        f32 = lambda t: t.detach().contiguous().float()
        idx = None if pb.img_indices is None else pb.img_indices.detach().reshape(-1).contiguous().long()
        delta, pl_delta = self.view_deltas() if idx is not None else (None, None)
        # No per-batch host-side range check of idx (a training batch costs no host sync): the kernels treat a view index outside
        # [0, ncam) as "no refinement" and scatter nothing for it in the adjoint, so a bad index cannot read or write out of range.
        # The reference raises IndexError when it indexes the delta tables with such an index (camera/ray_generator.py:108-121) - a
        # dataset / checkpoint view-count mismatch must not train silently with unrefined poses - so the FIRST bundle of every
        # generator is validated (one sync, outside a graph capture), and validate_view_indices() is there for the data loader.
        if idx is not None and (delta is not None or pl_delta is not None) and not self.__dict__.get("_idx_validated", False) \
                and not torch.cuda.is_current_stream_capturing():
            self.validate_view_indices(idx)
            self.__dict__["_idx_validated"] = True
        cc = lambda t: None if t is None else t.contiguous().float()
        poses = f32(pb.poses)
        o, d, p, near, far = _RaysIndexedHip.apply(cc(delta), cc(pl_delta), idx, f32(pb.h_indices).reshape(-1),
                                                   f32(pb.w_indices).reshape(-1), poses, f32(pb.pls),
                                                   (float(cam.cx), float(cam.cy), float(cam.fx), float(cam.fy),
                                                    bool(cfg.override_near_far_from_sphere), self.zn, self.zf))
        return RayBundle(origins=o, directions=d, pl_positions=p, nears=near, fars=far)

    def forward(self, pixel_bundle: RawPixelBundle) -> RayBundle:
        cam, cfg = self.camera, self.config
        if pixel_bundle.h_indices.is_cuda:
            return self._forward_hip(pixel_bundle)
        x = pixel_bundle.w_indices[..., 0] + 0.5        # pixel centres (:79-80)
        y = pixel_bundle.h_indices[..., 0] + 0.5
        idx = None if pixel_bundle.img_indices is None else pixel_bundle.img_indices[..., 0]
        dirs_cam = torch.stack([(x - cam.cx) / cam.fx, -(y - cam.cy) / cam.fy, -torch.ones_like(x)], dim=-1)
        R, t = pixel_bundle.poses[:, :3, :3], pixel_bundle.poses[:, :3, 3:]
        pls = pixel_bundle.pls
        if idx is not None:     # novel / video views have no training-view index and get no refinement (:103-105)
            if hasattr(self, "cam_pose_noise"):
                R, t = self._compose(self.cam_pose_noise[idx], R, t)
            if cfg.cam_opt_mode == "SO3xR3":
                R, t = self._compose(exp_map_SO3xR3(self.cam_pose_adjustment[idx]), R, t)
            elif cfg.cam_opt_mode == "SE3":
                R, t = self._compose(exp_map_SE3(self.cam_pose_adjustment[idx]), R, t)
            if hasattr(self, "pl_noise"):
                pls = pls + self.pl_noise[idx]
            if cfg.pl_opt:
                pls = pls + self.pl_adjustment[idx]
        d = F.normalize((dirs_cam[..., None, :] * R).sum(-1), dim=-1, p=2)      # R @ dir, unit length (:129-130)
        o = t[..., 0]
        if cfg.override_near_far_from_sphere:
            # mid-point of the chord through the unit sphere, -/+ 1 (:135-139)
            a = (d * d).sum(-1, keepdim=True)
            b = 2.0 * (o * d).sum(-1, keepdim=True)
            mid = 0.5 * (-b) / a
            near, far = mid - 1.0, mid + 1.0
        else:
            near, far = self.zn * torch.ones_like(o[..., :1]), self.zf * torch.ones_like(o[..., :1])
        return RayBundle(origins=o, directions=d, pl_positions=pls, nears=near, fars=far)
