"""Configuration dataclasses of the hot path - the shape/behaviour contract of the reference's defaults.

Mirrors the field names and default values of ``SDFNetConfig`` (fields/sdf_field.py:11-36),
``ReflectanceNetConfig`` (fields/reflectance_network.py:9-22), ``SingleVarianceNetConfig`` /
``NeuSRendererConfig`` / ``NeuSModelConfig`` (models/neus_hint_model.py:96-213) so that a reference config tree
can be passed field for field.  Branches the HIP path does not implement are rejected loudly by
``NeuSHintRenderer`` (see ``unsupported_reason``), never silently mis-computed.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from enum import Enum
from typing import List, Optional


class DepthComputationType(Enum):
    AlphaBlend = "alpha_blending"
    MaximalWeightPoint = "maximum_point"
    SphereTracing = "sphere_tracing"


class NormalComputationType(Enum):
    Analytic = "analytic"
    NormalizedAnalytic = "normalized_analytic"


@dataclass(frozen=True)
class SDFNetConfig:
    d_in: int = 3
    d_out_feat: int = 256
    d_hidden: int = 256
    n_layers: int = 8
    skip_in: List[int] = field(default_factory=lambda: [4])
    multi_res: int = 6
    init_bias: float = 0.5
    scale: float = 3.0
    geometric_init: bool = True
    weight_norm: bool = True
    inside_outside: bool = False


@dataclass(frozen=True)
class ReflectanceNetConfig:
    d_hidden: int = 256
    n_layers: int = 4
    weight_norm: bool = True
    multi_res: int = 4
    squeeze_out: bool = True


@dataclass(frozen=True)
class NeRFConfig:
    """The outside-NeRF background network (fields/nerf_density_field.py:12-25)."""
    d_hidden: int = 256
    n_layers: int = 8
    multi_res: int = 10
    multi_res_view: int = 4
    skips: List[int] = field(default_factory=lambda: [4])


@dataclass(frozen=True)
class SingleVarianceNetConfig:
    init_val: float = 0.3


@dataclass(frozen=True)
class NeuSRendererConfig:
    use_outside_nerf: bool = False
    n_samples: int = 64
    n_importance_samples: int = 64
    n_outside_samples: int = 32
    normal_type: NormalComputationType = NormalComputationType.NormalizedAnalytic
    up_sample_steps: int = 4
    depth_type: DepthComputationType = DepthComputationType.AlphaBlend
    shadow_hint: bool = True
    force_shadow_map: bool = False
    specular_hint: bool = True
    force_specular_cue: bool = False
    shadow_ray_offset: float = 1e-2
    specular_roughness: List[float] = field(default_factory=lambda: [0.02, 0.05, 0.13, 0.34])
    shadow_hint_gradient: bool = False
    specular_hint_gradient: bool = False
    n_shadow_importance_clip: int = -1
    n_shadow_samples: int = 64
    n_shadow_importance_samples: int = 64
    override_near_far_to_sphere: bool = True


@dataclass(frozen=True)
class NeuSModelConfig:
    sdf_network: SDFNetConfig = field(default_factory=SDFNetConfig)
    outside_nerf: NeRFConfig = field(default_factory=NeRFConfig)
    deviation_network: SingleVarianceNetConfig = field(default_factory=SingleVarianceNetConfig)
    reflectance_network: ReflectanceNetConfig = field(default_factory=ReflectanceNetConfig)
    renderer: NeuSRendererConfig = field(default_factory=NeuSRendererConfig)
    igr_weight: float = 0.1
    lr: float = 5e-4
    lr_alpha: float = 0.05
    warm_up_end: int = 5_000
    end_iter: int = 1_000_000
    anneal_end: int = 50_000
    geometry_warmup_end: int = 0
    batch_size: int = 512
    shadow_mini_chunk_size: int = 2048
    training_chunk_size: int = 512
    inference_chunk_size: int = 512


def sample_counts(r: "NeuSRendererConfig"):
    """(n_coarse, n_steps, n_new, s_coarse, s_new) the kernels run for this renderer config, or None if they cannot
    (models/neus_hint_model.py:696-713: n_importance_samples // up_sample_steps new samples per step on the primary ray; :373-412: the
    shadow march always takes 4 steps of n_shadow_importance_samples // 4)."""
    nc, ni, st = int(r.n_samples), int(r.n_importance_samples), int(r.up_sample_steps)
    snc, sni = int(r.n_shadow_samples), int(r.n_shadow_importance_samples)
    if ni <= 0 or st <= 0:
        st, nn = 0, 16
    else:
        nn = ni // st
    sn = sni // 4 if sni > 0 else 0
    ok = (2 <= nc <= 128 and 0 <= st <= 8 and 1 <= nn <= 16 and nc + st * nn <= 128 and 2 <= snc <= 64 and 0 <= sn <= 16
          and (sni <= 0 or sn >= 1) and snc + 4 * sn <= 128)
    return (nc, st, nn, snc, sn) if ok else None


def unsupported_reason(cfg: NeuSModelConfig) -> Optional[str]:
    """None if ``cfg`` is the network/renderer shape the gfx950 kernels are compiled for, else why not."""
    s, c, r = cfg.sdf_network, cfg.reflectance_network, cfg.renderer
    n = getattr(cfg, "outside_nerf", None) or NeRFConfig()
    checks = [
        # depth, skip position, input scale and weight-norm are compiled in; WIDTHS and encoding resolutions are upper bounds - a
        # narrower network runs zero-padded on the same kernels (packing.pad_to_compiled: exact, at the full network's cost)
        (s.d_in == 3 and s.n_layers == 8 and list(s.skip_in) == [4] and s.weight_norm and float(s.scale) == 3.0,
         "sdf_network must be an 8-layer / skip_in=[4] / scale=3 weight-normalised MLP (init_bias, geometric_init and "
         "inside_outside only choose the initial weights and are free)"),
        (1 <= s.multi_res <= 6 and 3 + 6 * s.multi_res < s.d_hidden <= 256 and s.d_hidden - (3 + 6 * s.multi_res) <= 217
         and 1 <= s.d_out_feat <= 256,
         "sdf_network: multi_res in 1..6, d_hidden <= 256 with d_hidden - (3 + 6 multi_res) in 1..217 (the skip layer's rows), "
         "d_out_feat in 1..256 (narrower than the compiled 256 / 6 / 256 runs zero-padded; wider is not built)"),
        (c.n_layers == 4 and c.weight_norm and c.squeeze_out and 1 <= c.d_hidden <= 256 and 1 <= c.multi_res <= 4,
         "reflectance_network must be a 4-layer weight-normalised sigmoid MLP with d_hidden <= 256 and multi_res in 1..4"),
        # (hint gradients together with the background run on the fused training step only, train_fused.py; forward() under autograd
        # refuses that pair at call time)
        (not r.use_outside_nerf or (r.n_outside_samples == 32 and r.n_importance_samples == 64 and r.n_shadow_importance_clip == -1),
         "use_outside_nerf needs n_outside_samples = 32, the 128-sample layout and the hit-point shadow mode"),
        (not r.use_outside_nerf or (n.d_hidden == 256 and n.n_layers == 8 and n.multi_res == 10 and n.multi_res_view == 4
                                    and list(n.skips) == [4]),
         "outside_nerf must be the default 8x256 / multires 10 + 4 / skips=[4] network"),
        (sample_counts(r) is not None,
         "sample counts: 2 <= n_samples, n_importance_samples // up_sample_steps in 1..16 (or n_importance_samples = 0), up_sample_steps <= 8 "
         "and n_samples + up_sample_steps * (n_importance_samples // up_sample_steps) <= 128; 2 <= n_shadow_samples <= 64, "
         "n_shadow_importance_samples // 4 in 1..16 (or 0) and n_shadow_samples + 4 * (n_shadow_importance_samples // 4) <= 128 "
         "(the kernels keep 128 slots per ray and draw at most 16 importance samples per step)"),
        (not ((r.use_outside_nerf or (r.shadow_hint and r.n_shadow_importance_clip > 0)) and sample_counts(r) != (64, 4, 16, 64, 16)),
         "the outside NeRF and the partial visibility hint need the default sample counts (64 + 64 / 4 steps, 64 + 64 on the shadow ray)"),
        (r.n_shadow_importance_clip in (-1, 1, 2, 4, 8, 16) or not r.shadow_hint,
         "n_shadow_importance_clip must be -1 (hit point) or 1, 2, 4, 8, 16 (partial visibility hint: that many shadow rays per ray)"),
        # force_* only adds the hint to has_*_hint (models/neus_hint_model.py:239-240): a no-op when the hint is on; with the hint
        # off the reference itself cannot run - the reflectance net's first layer is sized for the hint (:246-250) but built
        # without it (:256-257), so ReflectanceNetwork.forward never appends it (fields/reflectance_network.py:82-86) and lin0
        # fails on the shape
        (not (r.force_shadow_map and not r.shadow_hint),
         "force_shadow_map without shadow_hint fails in the reference itself (reflectance lin0 is sized for a visibility input "
         "that ReflectanceNetwork.forward never appends); enable shadow_hint"),
        (not (r.force_specular_cue and not r.specular_hint),
         "force_specular_cue without specular_hint fails in the reference itself (same lin0 shape mismatch); enable specular_hint"),
        # the VALUES of these two are kernel constants (NrhNet.specular_roughness / shadow_ray_offset); only the NUMBER of roughness
        # values is a shape: it sizes the reflectance net's first layer (models/neus_hint_model.py:246-250)
        (len(list(r.specular_roughness)) == 4 or not r.specular_hint, "specular_roughness must hold 4 values (the cue input is 4 x 9 wide)"),
        (all(float(x) > 0.0 for x in r.specular_roughness), "specular_roughness values must be positive"),
        (0.0 <= float(r.shadow_ray_offset) < 1.0, "shadow_ray_offset must lie in [0, 1)"),
    ]
    for ok, why in checks:
        if not ok:
            return why
    return None
