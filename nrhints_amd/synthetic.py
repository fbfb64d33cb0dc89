"""Synthetic rays / scenes for tests and the benchmark (numpy only, no device code).

The reference ships no data and the datasets are not reachable offline, so the
metric is quoted on synthetic random-weight scenes:

* rays: cameras on a radius-3 shell looking at the unit sphere, one point light
  per ray on a radius-4 shell, near/far from the unit-sphere intersection exactly
  as the reference's ray generator computes them (camera/ray_generator.py:135-139);
* scene "a": the reference's own initialisation (committed fixture);
* scene "b": a deterministic perturbation of scene "a" defined here, so that the
  fixture and every consumer derive it identically from one weight file.
"""
from __future__ import annotations

import numpy as np


def _unit(v: np.ndarray) -> np.ndarray:
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def make_rays(n: int, seed: int = 0, spread: float = 0.05):
    """Return (origins[n,3], directions[n,3], pl_positions[n,3], nears[n,1], fars[n,1]) float32.

    ``spread`` is the std of the jitter added to the direction that points at the
    origin: 0.05 -> nearly every ray crosses the unit sphere, 0.15 -> about half miss.
    """
    rs = np.random.RandomState(seed)
    o = 3.0 * _unit(rs.randn(n, 3))
    d = _unit(-_unit(o) + spread * rs.randn(n, 3))
    pl = 4.0 * _unit(rs.randn(n, 3))
    o, d, pl = (a.astype(np.float32) for a in (o, d, pl))
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    # near/far: mid -/+ 1 with mid = -(o.d)/(d.d)  (camera/ray_generator.py:135-139)
    a = np.sum(d * d, axis=-1, keepdims=True, dtype=np.float32)
    b = np.float32(2.0) * np.sum(o * d, axis=-1, keepdims=True, dtype=np.float32)
    mid = np.float32(0.5) * (-b) / a
    near = (mid - np.float32(1.0)).astype(np.float32)
    far = (mid + np.float32(1.0)).astype(np.float32)
    return o, d, pl, near, far


def make_image_rays(h: int, w: int, radius: float = 4.0, light_radius: float = 4.5, focal: float = 1111.1,
                    azimuth: float = 0.6, elevation: float = 0.5, row0: int = 0, row1: int | None = None):
    """Pin-hole camera on an orbit looking at the origin; one point light for the whole image.

    Pixel -> ray follows camera/ray_generator.py:79-139 (x right, y down, camera looks along -z,
    ``dirs = ((x-cx)/fx, -(y-cy)/fy, -1)``, rows [row0,row1) only so ranks can shard by row block).
    """
    row1 = h if row1 is None else row1
    ce, se, ca, sa = np.cos(elevation), np.sin(elevation), np.cos(azimuth), np.sin(azimuth)
    pos = radius * np.array([ce * ca, ce * sa, se])
    fwd = -pos / np.linalg.norm(pos)
    right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
    right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    rot = np.stack([right, up, -fwd], axis=1)  # columns = camera x, y, z axes in world
    ys, xs = np.meshgrid(np.arange(row0, row1) + 0.5, np.arange(w) + 0.5, indexing="ij")
    dirs = np.stack([(xs - w / 2) / focal, -(ys - h / 2) / focal, -np.ones_like(xs)], axis=-1).reshape(-1, 3)
    d = _unit(dirs @ rot.T).astype(np.float32)
    n = d.shape[0]
    o = np.broadcast_to(pos.astype(np.float32), (n, 3)).copy()
    lp = light_radius * np.array([np.cos(0.9) * np.cos(azimuth + 0.7), np.cos(0.9) * np.sin(azimuth + 0.7), np.sin(0.9)])
    pl = np.broadcast_to(lp.astype(np.float32), (n, 3)).copy()
    b = np.float32(2.0) * np.sum(o * d, axis=-1, keepdims=True, dtype=np.float32)
    a = np.sum(d * d, axis=-1, keepdims=True, dtype=np.float32)
    mid = np.float32(0.5) * (-b) / a
    return o, d, pl, (mid - 1).astype(np.float32), (mid + 1).astype(np.float32)


def perturb_state(state: dict, seed: int = 42, sigma: float = 0.02, sigma_pe: float = 0.004,
                  variance: float = 0.7, pe_cols: int = 36) -> dict:
    """Scene "b": break the sphere symmetry of the geometric init, deterministically.

    Adds N(0, sigma) to every SDF-trunk ``weight_v`` entry (N(0, sigma_pe) on the columns that read the
    positional-encoding part of the embedding, which the init leaves at exactly zero), small noise to the
    trunk biases, and sets the NeuS sharpness parameter to ``variance`` (inv_s = exp(10*variance) ~ 1.1e3).
    The reflectance net keeps its reference init.  ``pe_cols``: sinusoid columns of the embedding, 6 x multi_res (36 by default).
    """
    rs = np.random.RandomState(seed)
    out = {}
    for key in sorted(state.keys()):
        v = np.array(state[key], dtype=np.float32, copy=True)
        if key.startswith("sdf_network.lin") and key.endswith("weight_v"):
            noise = rs.randn(*v.shape).astype(np.float32)
            scale = np.full(v.shape, sigma, dtype=np.float32)
            layer = int(key.split(".")[1][3:])
            if layer == 0:
                scale[:, 3:] = sigma_pe
            if layer == 4:
                scale[:, -pe_cols:] = sigma_pe
            v = v + scale * noise
        elif key.startswith("sdf_network.lin") and key.endswith("bias"):
            v = v + np.float32(0.01) * rs.randn(*v.shape).astype(np.float32)
        elif key == "deviation_network.variance":
            v = np.array(variance, dtype=np.float32)
        out[key] = v
    return out


def naive_state(state: dict) -> dict:
    """State dict of the `pl-naive` model (no shadow / specular hints: reflectance input 316 instead of 361 wide,
    configs/main_config.py:67-76) derived deterministically from a full state: the hint columns of the first
    reflectance layer are dropped."""
    out = {k: np.array(v, copy=True) for k, v in state.items()}
    out["color_network.lin0.weight_v"] = out["color_network.lin0.weight_v"][:, :316].copy()
    return out


def one_hint_state(state: dict, shadow: bool) -> dict:
    """State dict of a model with ONE hint (shadow_hint xor specular_hint), derived from a full nr-hints state: the other hint's
    input columns of the first reflectance layer are dropped (fields/reflectance_network.py:44-52: [.., feature 316 | visibility
    encoding 9 | cue encoding 36])."""
    out = {k: np.array(v, copy=True) for k, v in state.items()}
    v = out["color_network.lin0.weight_v"]
    out["color_network.lin0.weight_v"] = (v[:, :325] if shadow else np.concatenate([v[:, :316], v[:, 325:]], axis=1)).copy()
    return out


def psnr(a, b) -> float:
    """10*log10(1/MSE), data range 1 (utils/metrics.py:8-9 via torchmetrics)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    mse = float(np.mean((a - b) ** 2))
    return float("inf") if mse == 0 else 10.0 * np.log10(1.0 / mse)
