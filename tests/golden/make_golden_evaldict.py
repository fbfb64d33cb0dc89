#!/usr/bin/env python3
"""Golden fixture for the evaluation loop, recorded by IMPORTING the reference's
``BaseNRHintPipeline.get_eval_dicts`` (pipelines/base_pipeline.py:93-160) and running it on one 24 x 32 view of scene b.
Build container only; writes tests/golden/evaldict_b.npz - plain data: the camera, the pose, the light, a synthetic ground
truth, and the three dictionaries the reference returns, plus the uint8 images its trainer writes from them
(trainer/trainer.py:343-352).

    python tests/golden/make_golden_evaldict.py

Stubs (modules the image lacks; none of them is on the measured path): jaxtyping (annotations), mcubes / imageio / cv2 (mesh and
file output), tyro (CLI), lpips (a network download) and torchmetrics.functional.image, whose ``peak_signal_noise_ratio`` is
restated from its published definition 10 log10(data_range^2 / MSE); SSIM and LPIPS are out of scope and the
stubs return NaN for them.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def _install_stubs():
    import torch

    class _Sub:
        def __getitem__(self, item):
            return object

    jt = types.ModuleType("jaxtyping")
    for name in ("Float", "Int", "Shaped", "Bool"):
        setattr(jt, name, _Sub())
    sys.modules["jaxtyping"] = jt
    for name in ("mcubes", "imageio", "cv2"):
        sys.modules.setdefault(name, types.ModuleType(name))
    tyro = types.ModuleType("tyro")
    tyro.conf = types.ModuleType("tyro.conf")
    tyro.conf.FlagConversionOff = _Sub()
    tyro.cli = lambda *a, **k: None
    sys.modules.setdefault("tyro", tyro)
    sys.modules.setdefault("tyro.conf", tyro.conf)

    tm = types.ModuleType("torchmetrics")
    tmf = types.ModuleType("torchmetrics.functional")
    tmi = types.ModuleType("torchmetrics.functional.image")

    def peak_signal_noise_ratio(preds, target, data_range=1.0):
        return 10.0 * torch.log10(torch.as_tensor(float(data_range) ** 2) / torch.mean((preds - target) ** 2))

    tmi.peak_signal_noise_ratio = peak_signal_noise_ratio
    tmi.structural_similarity_index_measure = lambda *a, **k: torch.tensor(float("nan"))
    tm.functional, tmf.image = tmf, tmi
    sys.modules.update({"torchmetrics": tm, "torchmetrics.functional": tmf, "torchmetrics.functional.image": tmi})

    lp = types.ModuleType("lpips")

    class _LPIPS:
        def __init__(self, *a, **k):
            pass

        def cpu(self):
            return self

        def __call__(self, *a, **k):
            return torch.tensor(float("nan"))

    lp.LPIPS = _LPIPS
    sys.modules["lpips"] = lp


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)
    import torch

    torch.set_num_threads(8)
    from camera.camera_model import CameraModel
    from configs.main_config import SystemConfig
    from data.data_loader import RawPixelBundle
    from data.shm_helper import NRDataSHMInfo
    from pipelines.base_pipeline import BaseNRHintPipeline

    from nrhints_amd.synthetic import perturb_state

    H, W = 24, 32
    cam = CameraModel(H=H, W=W, cx=15.5, cy=12.25, fx=44.0, fy=43.0, zn=0.1, zf=10.0)
    info = NRDataSHMInfo(total_image_num=1, num_image_per_split=[1, 0, 0], camera=cam, imgs_shm_name="", poses_shm_name="",
                         pls_shm_name="")
    torch.manual_seed(0)
    pipe = BaseNRHintPipeline(SystemConfig(), info).eval()
    state_a = dict(np.load(os.path.join(HERE, "scene_a_state.npz")))
    pipe.renderer.load_state_dict({k: torch.from_numpy(v) for k, v in perturb_state(state_a).items()})

    # camera on an orbit, looking at the origin (camera x right, y up, looks along -z: camera/ray_generator.py:100-110)
    az, el, radius = 0.6, 0.5, 4.0
    pos = radius * np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
    fwd = -pos / np.linalg.norm(pos)
    right = np.cross(fwd, [0.0, 0.0, 1.0])
    right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = np.stack([right, up, -fwd], axis=1)
    pose[:3, 3] = pos
    pl = (4.5 * np.array([np.cos(0.9) * np.cos(az + 0.7), np.cos(0.9) * np.sin(az + 0.7), np.sin(0.9)])).astype(np.float32)
    g = torch.Generator().manual_seed(11)
    rgb_gt = torch.rand(H, W, 3, generator=g)

    w_idx, h_idx = torch.meshgrid(torch.linspace(0, W - 1, W), torch.linspace(0, H - 1, H), indexing="xy")  # data_loader.py:206
    bundle = RawPixelBundle(img_indices=None, h_indices=h_idx[..., None], w_indices=w_idx[..., None], rgb_gt=rgb_gt,
                            poses=torch.from_numpy(pose)[None, None].repeat(H, W, 1, 1),
                            pls=torch.from_numpy(pl)[None, None].repeat(H, W, 1))
    img, metrics, tensors = pipe.get_eval_dicts(bundle, torch.device("cpu"))

    out = {"camera": np.array([H, W, cam.cx, cam.cy, cam.fx, cam.fy], dtype=np.float64), "pose": pose, "pl": pl,
           "rgb_gt": rgb_gt.numpy(), "psnr": np.float64(metrics["psnr"])}
    for k, v in img.items():
        out["img." + k] = np.asarray(v)
    for k, v in tensors.items():
        out["tensor." + k] = np.asarray(v)
    # what the trainer writes to disk from img_dict (trainer/trainer.py:343-352)
    for k, v in img.items():
        v = np.asarray(v)
        if "normal" in k:
            v = v * 0.5 + 0.5
        if v.shape[-1] == 1:
            v = v[..., 0]
        out["u8." + k] = (v * 255).clip(0, 255).astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "evaldict_b.npz"), **out)
    for k, v in out.items():
        print(k, getattr(v, "shape", None), getattr(v, "dtype", None))


if __name__ == "__main__":
    main()
