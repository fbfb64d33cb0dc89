"""nrhints_amd - MI355X-native (gfx950) implementation of the NRHints volumetric-rendering hot path.

Public surface mirrors the reference's hot-path names (models/neus_hint_model.py, camera/ray_utils.py):
``NeuSHintRenderer``, ``NeuSModelConfig`` (+ sub-configs), ``RayBundle``, ``RenderOutput``, ``td_concat``.
Importing the package never loads the HIP library; the first rendering call does, and fails loudly if
it is missing (``nrhints_amd._lib.HipExtensionMissing``).
"""
from .config import (DepthComputationType, NeRFConfig, NeuSModelConfig, NeuSRendererConfig, NormalComputationType,
                     ReflectanceNetConfig, SDFNetConfig, SingleVarianceNetConfig, unsupported_reason)
from .containers import RawPixelBundle, RayBundle, RenderOutput, td_concat
from .ray_generator import RayGenerator, RayGeneratorConfig
from .renderer import NeuSHintRenderer

__all__ = ["NeuSHintRenderer", "NeuSModelConfig", "NeuSRendererConfig", "SDFNetConfig", "ReflectanceNetConfig",
           "SingleVarianceNetConfig", "NeRFConfig", "DepthComputationType", "NormalComputationType", "RayBundle", "RenderOutput",
           "td_concat", "unsupported_reason", "RawPixelBundle", "RayGenerator", "RayGeneratorConfig"]
__version__ = "0.1.0"
