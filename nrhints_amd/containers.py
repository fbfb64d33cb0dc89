"""Typed tensor containers crossing the renderer boundary.

Own, minimal counterparts of the reference's ``RayBundle`` (camera/ray_utils.py:214-235), ``RenderOutput``
(models/neus_hint_model.py:216-233) and of the batch semantics they inherit from ``TensorDataclass``
(utils/tensor_dataclass.py:29-334): every tensor field shares leading *batch* dimensions; fields listed in
``_trailing`` keep that many trailing dimensions (default 1).  Supported: ``.shape``, ``len``, indexing,
``reshape``, ``flatten``, ``to``, ``detach`` and ``td_concat`` - what pipelines/base_pipeline.py:107-133 uses.
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass
from typing import ClassVar, Dict, Optional, Sequence

import torch


class TensorBatch:
    _trailing: ClassVar[Dict[str, int]] = {}

    # -- helpers ---------------------------------------------------------------------------------
    def _tensor_items(self):
        for f in dataclasses.fields(self):
            v = getattr(self, f.name)
            if isinstance(v, torch.Tensor):
                yield f.name, v, self._trailing.get(f.name, 1)

    def __post_init__(self):
        shapes = [tuple(v.shape[: v.dim() - k]) for _, v, k in self._tensor_items()]
        if not shapes:
            raise ValueError(f"{type(self).__name__} needs at least one tensor field")
        batch = torch.broadcast_shapes(*shapes)
        for name, v, k in list(self._tensor_items()):
            if tuple(v.shape[: v.dim() - k]) != tuple(batch):
                setattr(self, name, v.broadcast_to(tuple(batch) + tuple(v.shape[v.dim() - k:])))
        object.__setattr__(self, "_shape", tuple(batch))

    def _map(self, fn):
        kw = {}
        for f in dataclasses.fields(self):
            v = getattr(self, f.name)
            kw[f.name] = fn(v, self._trailing.get(f.name, 1)) if isinstance(v, torch.Tensor) else v
        return type(self)(**kw)

    # -- public surface --------------------------------------------------------------------------
    @property
    def shape(self):
        return self._shape

    def __len__(self):
        if not self._shape:
            raise TypeError("len() of a 0-d TensorBatch")
        return self._shape[0]

    @property
    def size(self):
        n = 1
        for s in self._shape:
            n *= s
        return n

    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        return self._map(lambda v, k: v[idx + (slice(None),) * k])

    def reshape(self, shape: Sequence[int]):
        shape = tuple(shape) if not isinstance(shape, int) else (shape,)
        return self._map(lambda v, k: v.reshape(shape + tuple(v.shape[v.dim() - k:])))

    def flatten(self):
        return self.reshape((-1,))

    def to(self, device):
        return self._map(lambda v, k: v.to(device))

    def detach(self):
        return self._map(lambda v, k: v.detach())

    def cpu(self):
        return self.to("cpu")


def td_concat(items, dim: int = 0):
    """Concatenate along a batch dimension (utils/tensor_dataclass.py:337-357)."""
    first = items[0]
    kw = {}
    for f in dataclasses.fields(first):
        v = getattr(first, f.name)
        kw[f.name] = torch.cat([getattr(it, f.name) for it in items], dim=dim) if isinstance(v, torch.Tensor) else v
    return type(first)(**kw)


@dataclass
class RayBundle(TensorBatch):
    """Input of ``NeuSHintRenderer.forward`` (produced by camera/ray_generator.py:144-150)."""
    origins: torch.Tensor                    # [*bs,3]
    directions: torch.Tensor                 # [*bs,3] unit length
    pl_positions: torch.Tensor               # [*bs,3] point-light position per ray
    camera_indices: Optional[torch.Tensor] = None
    pixel_area: Optional[torch.Tensor] = None
    nears: Optional[torch.Tensor] = None     # [*bs,1]
    fars: Optional[torch.Tensor] = None      # [*bs,1]
    metadata: Optional[dict] = None


@dataclass
class RenderOutput(TensorBatch):
    """Output of ``NeuSHintRenderer.forward`` (models/neus_hint_model.py:739-750)."""
    rgb: torch.Tensor                           # [*bs,3]
    depth: torch.Tensor                         # [*bs,1]
    weights: torch.Tensor                       # [*bs,128]
    s_val: torch.Tensor                         # [*bs,128]  (annotated "bs 1" upstream, actually per sample)
    inside_sphere: torch.Tensor                 # [*bs,128]
    relax_inside_sphere: torch.Tensor           # [*bs,128]  (= inside_sphere, upstream quirk at :745)
    analytic_normals: torch.Tensor              # [*bs,128,3]
    normalized_analytic_normals: torch.Tensor   # [*bs,128,3]
    visibilities: Optional[torch.Tensor] = None  # [*bs,1]
    specular_cue: Optional[torch.Tensor] = None  # [*bs,128,4]

    _trailing: ClassVar[Dict[str, int]] = {"analytic_normals": 2, "normalized_analytic_normals": 2, "specular_cue": 2}


@dataclass
class RawPixelBundle(TensorBatch):
    """Pixels before ray generation (data/data_loader.py:79-88): what the data loader hands to the ray generator."""
    img_indices: Optional[torch.Tensor]      # [*bs,1] int64 training-view index, None for video / novel views
    h_indices: torch.Tensor                  # [*bs,1] pixel row
    w_indices: torch.Tensor                  # [*bs,1] pixel column
    poses: torch.Tensor                      # [*bs,4,4] camera-to-world
    pls: torch.Tensor                        # [*bs,3] point-light position
    rgb_gt: Optional[torch.Tensor] = None    # [*bs,3]

    _trailing: ClassVar[Dict[str, int]] = {"poses": 2}
