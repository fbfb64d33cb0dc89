"""Ray sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

Rays are independent units of the hot path: nothing is exchanged inside ``NeuSHintRenderer.forward`` and the 5.6 MB of
weights are replicated, so multi-GPU rendering needs exactly one collective - gathering the pixels.

* BASELINE config 4 (one image over N ranks): ``render_sharded`` gives every rank a contiguous slab of the flattened
  ray list (row blocks of the image), renders it locally and all-gathers ``rgb`` (and optionally depth / visibility);
  at 800x800 that is 0.96 MB per rank, latency-bound over xGMI and far below one chunk of compute.
* The reference instead assigns whole views to ranks (trainer/trainer.py:288-296); ``views_of_rank`` reproduces that
  split for benchmark / evaluation drivers (no collective at all on the data path).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .containers import RayBundle


def slab_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slab [lo, hi) of ``n`` rays for ``rank``: sizes differ by at most one, order preserved."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def views_of_rank(n_views: int, rank: int, world: int, skip: int = 1) -> List[int]:
    """The reference's view split: idx = rank*skip, step skip*world (trainer/trainer.py:288-296)."""
    return list(range(rank * skip, n_views, skip * world))


def render_sharded(render_fn: Callable[[RayBundle], "object"], rays: RayBundle,
                   fields: Sequence[str] = ("rgb",), group: Optional[dist.ProcessGroup] = None) -> Dict[str, torch.Tensor]:
    """Render ``rays`` (identical on every rank, batch shape [N]) with each rank doing one slab; returns the gathered
    ``fields`` ([N, C] each, identical on every rank).  ``render_fn`` is e.g. ``lambda rb: model(rb, background_rgb=bg)``.

    Works without an initialised process group (single process: plain local render)."""
    n = rays.shape[0]
    if not (dist.is_available() and dist.is_initialized()):
        out = render_fn(rays)
        return {f: getattr(out, f) for f in fields}
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = slab_bounds(n, rank, world)
    local = render_fn(rays[lo:hi])
    cap = slab_bounds(n, 0, world)[1]  # largest slab
    result = {}
    for f in fields:
        t = getattr(local, f)
        pad = torch.zeros((cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: hi - lo] = t
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
        sizes = [slab_bounds(n, r, world) for r in range(world)]
        result[f] = torch.cat([p[: b - a] for p, (a, b) in zip(parts, sizes)], dim=0)
    return result
