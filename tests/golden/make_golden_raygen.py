#!/usr/bin/env python3
"""Golden fixture for the ray generator with pose / light refinement, recorded by IMPORTING the
reference (camera/ray_generator.py:42-150, camera/lie_groups.py).  Build container only; writes tests/golden/raygen.npz
(plain data: inputs, expected rays, expected gradients of a fixed scalar w.r.t. the learnable deltas).

    python tests/golden/make_golden_raygen.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _install_stubs():
    class _Sub:
        def __getitem__(self, item):
            return object

    jt = types.ModuleType("jaxtyping")
    for name in ("Float", "Int", "Shaped", "Bool"):
        setattr(jt, name, _Sub())
    sys.modules["jaxtyping"] = jt
    for name in ("mcubes", "imageio", "cv2"):
        sys.modules.setdefault(name, types.ModuleType(name))
    # data/data_loader.py (home of RawPixelBundle) pulls in the CLI library through data/shm_helper.py
    tyro = types.ModuleType("tyro")
    tyro.conf = types.ModuleType("tyro.conf")
    tyro.conf.FlagConversionOff = _Sub()
    tyro.cli = lambda *a, **k: None
    sys.modules.setdefault("tyro", tyro)
    sys.modules.setdefault("tyro.conf", tyro.conf)


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    import torch
    from camera.camera_model import CameraModel
    from camera.lie_groups import exp_map_SE3, exp_map_SO3xR3
    from camera.ray_generator import RayGenerator, RayGeneratorConfig
    from data.data_loader import RawPixelBundle

    out = {}
    g = torch.Generator().manual_seed(7)
    # exponential maps, including the near-zero branches
    tv = torch.randn(12, 6, generator=g) * torch.tensor([0.3, 0.3, 0.3, 0.5, 0.5, 0.5])
    tv[0] = 0.0
    tv[1, 3:] = torch.tensor([1e-3, -2e-3, 5e-4])
    tv[2, 3:] = torch.tensor([3e-3, 4e-3, 0.0])
    out["tangent"] = tv.numpy()
    out["exp_SO3xR3"] = exp_map_SO3xR3(tv).numpy()
    out["exp_SE3"] = exp_map_SE3(tv).numpy()

    cam = CameraModel(H=48, W=64, cx=31.5, cy=24.25, fx=70.0, fy=68.0, zn=0.1, zf=10.0)
    out["camera"] = np.array([cam.H, cam.W, cam.cx, cam.cy, cam.fx, cam.fy, cam.zn, cam.zf], dtype=np.float64)
    ncam, n = 5, 40
    img = torch.randint(0, ncam, (n, 1), generator=g)
    hi = torch.randint(0, cam.H, (n, 1), generator=g).float()
    wi = torch.randint(0, cam.W, (n, 1), generator=g).float()
    # camera-to-world poses: random rotations (QR) at radius ~4
    q, _ = torch.linalg.qr(torch.randn(ncam, 3, 3, generator=g))
    q = q * torch.sign(torch.linalg.det(q))[:, None, None]
    pos = torch.nn.functional.normalize(torch.randn(ncam, 3, generator=g), dim=-1) * 4.0
    poses_cam = torch.eye(4).repeat(ncam, 1, 1)
    poses_cam[:, :3, :3] = q
    poses_cam[:, :3, 3] = pos
    poses = poses_cam[img[:, 0]]
    pls = (pos + 0.3 * torch.randn(ncam, 3, generator=g))[img[:, 0]]
    for k, v in (("img_indices", img), ("h_indices", hi), ("w_indices", wi), ("poses", poses), ("pls", pls)):
        out[k] = v.numpy()
    c3 = torch.randn(n, 3, generator=g)
    out["probe"] = c3.numpy()

    def run(tag, cfg, adj=None, pladj=None, seed=None, with_idx=True):
        if seed is not None:
            torch.manual_seed(seed)
        rg = RayGenerator(cam, ncam, cfg)
        if adj is not None:
            rg.cam_pose_adjustment.data.copy_(adj)
        if pladj is not None:
            rg.pl_adjustment.data.copy_(pladj)
        pb = RawPixelBundle(img_indices=img if with_idx else None, h_indices=hi, w_indices=wi, poses=poses, pls=pls, rgb_gt=None)
        rb = rg(pb)
        for k in ("origins", "directions", "pl_positions", "nears", "fars"):
            out[f"{tag}.{k}"] = getattr(rb, k).detach().numpy()
        for bname in ("cam_pose_noise", "pl_noise"):
            if hasattr(rg, bname):
                out[f"{tag}.{bname}"] = getattr(rg, bname).numpy()
        params = [p for p in rg.parameters()]
        if params and with_idx:
            loss = (rb.origins * c3).sum() + (rb.directions * c3.flip(0)).sum() * 2.0 + (rb.pl_positions * c3).sum() * 0.5 + \
                   (rb.nears * rb.fars).sum() * 0.1
            grads = torch.autograd.grad(loss, params)
            for (name, _), gr in zip(rg.named_parameters(), grads):
                out[f"{tag}.grad.{name}"] = gr.numpy()

    adj = torch.randn(ncam, 6, generator=g) * torch.tensor([0.05, 0.05, 0.05, 0.1, 0.1, 0.1])
    adj[0] = 0.0            # exactly-zero delta: clamp branch of the SO3 map
    pladj = torch.randn(ncam, 3, generator=g) * 0.1
    out["adj"], out["pladj"] = adj.numpy(), pladj.numpy()
    run("off", RayGeneratorConfig())
    run("so3", RayGeneratorConfig(cam_opt_mode="SO3xR3", pl_opt=True), adj, pladj)
    run("se3", RayGeneratorConfig(cam_opt_mode="SE3"), adj)
    run("video", RayGeneratorConfig(cam_opt_mode="SO3xR3", pl_opt=True), adj, pladj, with_idx=False)
    run("noise", RayGeneratorConfig(cam_opt_mode="SO3xR3", cam_position_noise_std=0.02, cam_orientation_noise_std=0.03,
                                    pl_position_noise_std=0.05), adj, seed=11)
    run("zplanes", RayGeneratorConfig(override_near_far_from_sphere=False))
    np.savez_compressed(os.path.join(HERE, "raygen.npz"), **out)
    print("wrote raygen.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
