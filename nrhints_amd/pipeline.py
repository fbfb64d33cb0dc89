"""Evaluation-loop counterpart for the hot path.

The reference renders a test view as 1 250 chunks of 512 rays, copies each chunk's full ``RenderOutput`` (6.7 KB/ray) to
the host, concatenates, and reduces the normal maps with an ``einsum`` on the CPU (pipelines/base_pipeline.py:107-133).
Here one view is: rays generated on the device from (pose, intrinsics, light) by a HIP kernel
(camera/ray_generator.py:79-139 without pose deltas), one pass through the renderer in 131 072-ray chunks, and the
per-pixel products - rgb, depth, shadow map, the two weighted normal maps - reduced inside the composite kernel, so a
frame leaves the GPU as 15 floats per pixel.  PSNR is ``10 log10(1 / MSE)`` (utils/metrics.py:8-9).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from . import _lib
from .containers import RayBundle


@dataclass(frozen=True)
class CameraModel:
    """Pin-hole intrinsics (camera/camera_model.py:5-15)."""
    H: int
    W: int
    cx: float
    cy: float
    fx: float
    fy: float


def generate_rays(camera: CameraModel, pose: torch.Tensor, pl: torch.Tensor, device, row0: int = 0,
                  row1: Optional[int] = None) -> RayBundle:
    """Rays of image rows [row0, row1) for one view: ``pose`` [3,4] or [4,4] camera-to-world, ``pl`` [3] light."""
    lib = _lib.load()
    row1 = camera.H if row1 is None else row1
    n = (row1 - row0) * camera.W
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)
    o, d, p, near, far = new(n, 3), new(n, 3), new(n, 3), new(n, 1), new(n, 1)
    pose_h = (ctypes.c_float * 12)(*[float(v) for v in pose.detach().cpu().reshape(-1)[:12].tolist()])
    pl_h = (ctypes.c_float * 3)(*[float(v) for v in pl.detach().cpu().reshape(-1)[:3].tolist()])
    P = _lib.ptr
    with torch.cuda.device(device):
        rc = lib.nrh_generate_rays(pose_h, pl_h, camera.cx, camera.cy, camera.fx, camera.fy, camera.W, row0, row1 - row0,
                                   P(o), P(d), P(p), P(near), P(far), _lib.stream_handle())
    _lib.check(rc, "nrh_generate_rays")
    return RayBundle(origins=o, directions=d, pl_positions=p, nears=near, fars=far)


@torch.no_grad()
def render_image(renderer, camera: CameraModel, pose: torch.Tensor, pl: torch.Tensor, white_background: bool = True,
                 rgb_gt: Optional[torch.Tensor] = None, row0: int = 0, row1: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """One evaluation view (or a row block of it).  Returns device tensors shaped [rows, W, C]:
    rgb, depth, shadow_map, analytic_normals / normalized_analytic_normals (rotated into the camera frame as
    pipelines/base_pipeline.py:123-131), and ``psnr`` (python float) if ``rgb_gt`` [rows, W, 3] is given."""
    device = next(renderer.parameters()).device
    row1 = camera.H if row1 is None else row1
    rays = generate_rays(camera, pose, pl, device, row0, row1)
    bg = torch.full((1, 3), 1.0 if white_background else 0.0, device=device)
    res = renderer.render_products(rays, bg)
    rows, W = row1 - row0, camera.W
    rot = torch.linalg.inv(pose.to(device=device, dtype=torch.float32)[:3, :3])
    to_cam = lambda m: (m @ rot.T).reshape(rows, W, 3)     # rot @ n per pixel
    out = {"rgb": res["rgb"].reshape(rows, W, 3), "depth": res["depth"].reshape(rows, W, 1),
           "shadow_map": res["visibilities"].reshape(rows, W, 1),
           "analytic_normals": to_cam(res["normal_map"]),
           "normalized_analytic_normals": to_cam(res["normalized_normal_map"])}
    if rgb_gt is not None:
        mse = torch.mean((out["rgb"] - rgb_gt.to(device)) ** 2)
        out["psnr"] = float(10.0 * torch.log10(1.0 / mse))
    return out
