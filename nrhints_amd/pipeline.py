"""Evaluation-loop counterpart for the hot path.

The reference renders a test view as 1 250 chunks of 512 rays, copies each chunk's full ``RenderOutput`` (6.7 KB/ray) to
the host, concatenates, and reduces the normal maps with an ``einsum`` on the CPU (pipelines/base_pipeline.py:107-133).
Here one view is: rays generated on the device from (pose, intrinsics, light) by a HIP kernel
(camera/ray_generator.py:79-139 without pose deltas), one pass through the renderer in 131 072-ray chunks, and the
per-pixel products - rgb, depth, shadow map, the two weighted normal maps - reduced inside the composite kernel, so a
frame leaves the GPU as 15 floats per pixel.  PSNR is ``10 log10(1 / MSE)`` (utils/metrics.py:8-9).

``get_eval_dicts`` returns the reference's three dictionaries (same keys, shapes and dtypes) for callers that consume them;
``to_uint8_images`` is the conversion its trainer applies before writing PNGs (trainer/trainer.py:343-352).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
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
                 rgb_gt: Optional[torch.Tensor] = None, row0: int = 0, row1: Optional[int] = None,
                 specular_hint: bool = False) -> Dict[str, torch.Tensor]:
    """One evaluation view (or a row block of it).  Returns device tensors shaped [rows, W, C]:
    rgb, depth, shadow_map, analytic_normals / normalized_analytic_normals (rotated into the camera frame as
    pipelines/base_pipeline.py:123-131), ``specular_hint`` [rows, W, 4] on request (the per-ray row of the reference's
    [rows, W, 128, 4] tensor), and ``psnr`` (python float) if ``rgb_gt`` [rows, W, 3] is given."""
    device = next(renderer.parameters()).device
    row1 = camera.H if row1 is None else row1
    rays = generate_rays(camera, pose, pl, device, row0, row1)
    bg = torch.full((1, 3), 1.0 if white_background else 0.0, device=device)
    res = renderer.render_products(rays, bg, specular_cue=specular_hint)
    rows, W = row1 - row0, camera.W
    # rot = inverse of the view's rotation, on the host as the reference does (:122); applied per pixel as three multiply-adds
    rot = torch.linalg.inv(pose.detach().to(device="cpu", dtype=torch.float32)[:3, :3]).to(device)
    to_cam = lambda m: (m[:, None, :] * rot[None, :, :]).sum(-1).reshape(rows, W, 3)     # rot @ n per pixel
    out = {"rgb": res["rgb"].reshape(rows, W, 3), "depth": res["depth"].reshape(rows, W, 1),
           "shadow_map": res["visibilities"].reshape(rows, W, 1),
           "analytic_normals": to_cam(res["normal_map"]),
           "normalized_analytic_normals": to_cam(res["normalized_normal_map"])}
    if "cue_ray" in res:
        out["specular_hint"] = res["cue_ray"].reshape(rows, W, 4)
    if rgb_gt is not None:
        mse = torch.mean((out["rgb"] - rgb_gt.to(device)) ** 2)
        out["psnr"] = float(10.0 * torch.log10(1.0 / mse))
    return out


def get_eval_dicts(renderer, camera: CameraModel, pose: torch.Tensor, pl: torch.Tensor,
                   rgb_gt: Optional[torch.Tensor] = None, white_background: bool = True,
                   specular_hint: bool = True) -> Tuple[Dict[str, np.ndarray], Dict[str, float], Dict[str, np.ndarray]]:
    """The reference's ``BaseNRHintPipeline.get_eval_dicts`` for one full view (pipelines/base_pipeline.py:93-160):
    ``(img_dict, metrics_dict, tensor_dict)`` of host numpy arrays with its keys and shapes -

      img_dict     rgb [H,W,3], analytic_normals [H,W,3], normalized_analytic_normals [H,W,3] (camera frame),
                   rgb_gt [H,W,3] if given, shadow_map [H,W,1] when the model has the shadow hint
      metrics_dict psnr (if ``rgb_gt``); SSIM / LPIPS are not computed here (third-party metric networks, SURVEY.md §8 out
                   of scope) - feed ``img_dict['rgb']`` to them as before
      tensor_dict  depth [H,W,1], specular_hint [H,W,128,4] when the model has the specular hint: a READ-ONLY broadcast view of
                   the per-ray [H,W,4] hint (the reference stores the same row 128 times; ``np.save`` writes it out in full)

    View registration (``register_view``, the 500 Adam steps on the ray-generator deltas when camera / light refinement is on)
    is the caller's job - it is a training loop over `RayGenerator` parameters, see training.py."""
    want_cue = specular_hint and bool(getattr(renderer, "has_specular_hint", False))
    out = render_image(renderer, camera, pose, pl, white_background=white_background, rgb_gt=rgb_gt, specular_hint=want_cue)
    host = lambda t: t.detach().cpu().numpy()
    img = {"rgb": host(out["rgb"]), "analytic_normals": host(out["analytic_normals"]),
           "normalized_analytic_normals": host(out["normalized_analytic_normals"])}
    if rgb_gt is not None:
        img["rgb_gt"] = host(rgb_gt)
    if getattr(renderer, "has_shadow_hint", False):
        img["shadow_map"] = host(out["shadow_map"])
    metrics = {"psnr": out["psnr"]} if rgb_gt is not None else {}
    tensors = {"depth": host(out["depth"])}
    if want_cue:
        cue = host(out["specular_hint"])
        tensors["specular_hint"] = np.broadcast_to(cue[:, :, None, :], cue.shape[:2] + (128, 4))
    return img, metrics, tensors


def to_uint8_images(img_dict: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """What the reference's trainer writes as PNGs from ``img_dict`` (trainer/trainer.py:343-352): normal maps go from
    [-1, 1] to [0, 1], a trailing axis of 1 is dropped, then ``(v * 255).clip(0, 255)`` truncated to uint8."""
    out = {}
    for k, v in img_dict.items():
        v = np.asarray(v)
        if "normal" in k:
            v = v * 0.5 + 0.5
        if v.shape[-1] == 1:
            v = v[..., 0]
        out[k] = (v * 255).clip(0, 255).astype(np.uint8)
    return out
