"""Ray sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

Rays are independent units of the hot path: nothing is exchanged inside ``NeuSHintRenderer.forward`` and the 5.6 MB of
weights are replicated, so multi-GPU rendering needs exactly one collective - gathering the pixels.

* BASELINE config 4 (one image over N ranks): every rank renders a contiguous slab of the flattened ray list (row blocks
  of the image) and ONE ``all_gather_into_tensor`` on a preallocated flat buffer brings the requested per-pixel fields of
  all slabs to every rank - at 800x800 and 8 ranks 0.96 MB per rank for ``rgb``, latency-bound over xGMI and far below
  one chunk of compute.  ``render_sharded`` takes the whole ray list (identical on every rank) and slices it;
  ``render_slab`` takes ONLY this rank's slab, for callers that can generate their rays per slab (``slab_bounds`` says
  which pixels): no rank then ever holds or generates the full bundle.
* The reference instead assigns whole views to ranks (trainer/trainer.py:288-296); ``views_of_rank`` reproduces that
  split for benchmark / evaluation drivers (no collective at all on the data path).
"""
from __future__ import annotations

import time
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


# (device, dtype, rows, columns, world) -> (send [cap, C], recv [world * cap, C]): allocated once, reused by every frame
_GATHER_BUFFERS: Dict[tuple, Tuple[torch.Tensor, torch.Tensor]] = {}


def _buffers(device, dtype, cap: int, cols: int, world: int):
    key = (str(device), dtype, cap, cols, world)
    b = _GATHER_BUFFERS.get(key)
    if b is None:
        if len(_GATHER_BUFFERS) > 8:        # a driver that keeps changing its frame size must not accumulate buffers
            _GATHER_BUFFERS.clear()
        b = _GATHER_BUFFERS[key] = (torch.zeros(cap, cols, dtype=dtype, device=device), torch.empty(world * cap, cols, dtype=dtype, device=device))
    return b


def render_slab(render_fn: Callable[[RayBundle], "object"], my_rays: RayBundle, n_total: int, fields: Sequence[str] = ("rgb",),
                group: Optional[dist.ProcessGroup] = None, stats: Optional[dict] = None) -> Dict[str, torch.Tensor]:
    """This rank renders ``my_rays`` - rays [lo, hi) = slab_bounds(n_total, rank, world) of a frame of ``n_total`` rays - and all
    ranks receive the ``fields`` of the whole frame ([n_total, C] each, identical on every rank; views of one gathered buffer
    when n_total divides evenly, which is what an 800x800 frame over 1 / 2 / 4 / 8 ranks does).

    One collective per frame: the fields are packed side by side into one [largest slab, sum C] buffer (padded with zeros where
    this rank's slab is one ray shorter) and gathered with a single ``all_gather_into_tensor`` into a preallocated
    [world * largest slab, sum C] buffer - no per-rank tensor lists, no per-field collectives.
    ``stats`` (optional dict): receives ``host_s`` - wall time this call spent on the host (enqueueing the render and the
    collective; nothing here synchronises the device)."""
    t0 = time.perf_counter()
    if not (dist.is_available() and dist.is_initialized()):
        out = render_fn(my_rays)
        if stats is not None:
            stats["host_s"] = stats.get("host_s", 0.0) + time.perf_counter() - t0
        return {f: getattr(out, f) for f in fields}
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = slab_bounds(n_total, rank, world)
    if my_rays.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} of {world} must bring rays [{lo}, {hi}) of {n_total}: {hi - lo} rays, got {my_rays.shape[0]}")
    local = render_fn(my_rays)
    parts = [getattr(local, f) for f in fields]
    widths = [int(torch.Size(t.shape[1:]).numel()) for t in parts]       # (an empty slab - fewer rays than ranks - still has its columns)
    cap = slab_bounds(n_total, 0, world)[1]          # largest slab (rank 0's)
    send, recv = _buffers(parts[0].device, parts[0].dtype, cap, sum(widths), world)
    c = 0
    for t, w in zip(parts, widths):
        send[: hi - lo, c: c + w].copy_(t.reshape(hi - lo, w))
        c += w
    dist.all_gather_into_tensor(recv, send, group=group)
    if n_total % world == 0:
        frame = recv                                   # slabs are contiguous and unpadded: the buffer IS the frame
    else:
        rem = n_total % world                          # ranks < rem hold ``cap`` rays, the others ``cap - 1`` + one padding row
        cols = recv.shape[1]
        body = recv.view(world, cap, cols)
        frame = torch.cat([body[:rem].reshape(rem * cap, cols), body[rem:, : cap - 1].reshape((world - rem) * (cap - 1), cols)], dim=0)
    result, c = {}, 0
    for f, t, w in zip(fields, parts, widths):
        result[f] = frame[:, c: c + w].reshape((n_total,) + tuple(t.shape[1:]))
        c += w
    if stats is not None:
        stats["host_s"] = stats.get("host_s", 0.0) + time.perf_counter() - t0
    return result


def render_sharded(render_fn: Callable[[RayBundle], "object"], rays: RayBundle, fields: Sequence[str] = ("rgb",),
                   group: Optional[dist.ProcessGroup] = None, stats: Optional[dict] = None) -> Dict[str, torch.Tensor]:
    """Render ``rays`` (identical on every rank, batch shape [N]) with each rank doing one slab; returns the gathered
    ``fields`` ([N, C] each, identical on every rank).  ``render_fn`` is e.g. ``lambda rb: model(rb, background_rgb=bg)``.
    Callers that can produce their slab alone should use ``render_slab`` (no rank needs the other ranks' rays).

    Works without an initialised process group (single process: plain local render).  The results alias a reused gather
    buffer: copy what must outlive the next call."""
    n = rays.shape[0]
    if not (dist.is_available() and dist.is_initialized()):
        return render_slab(render_fn, rays, n, fields, group, stats)
    lo, hi = slab_bounds(n, dist.get_rank(group), dist.get_world_size(group))
    return render_slab(render_fn, rays[lo:hi], n, fields, group, stats)
