"""The N > 1 path on CPU: world_size-2 and world_size-8 gloo processes shard rays, render their slab with a deterministic
stand-in renderer (the HIP library needs a GPU) and gather the pixels with ONE all_gather_into_tensor on a preallocated flat
buffer (nrhints_amd/parallel.py); the result must equal the single-process render - for frame sizes that divide evenly over
the ranks (the gathered buffer IS the frame), that do not (one padding row on the shorter slabs), and that are smaller than the
world (empty slabs).  ``render_slab``: every rank brings only its own slab."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import nrhints_amd as na
from nrhints_amd.parallel import render_sharded, render_slab, slab_bounds, views_of_rank
from nrhints_amd.synthetic import make_rays


def _fake_render(rb: na.RayBundle):
    """Any pure function of the rays will do: a cheap analytic shading so slabs are distinguishable."""
    o, d, pl = rb.origins, rb.directions, rb.pl_positions
    t = -(o * d).sum(-1, keepdim=True)
    p = o + d * t
    rgb = torch.sigmoid(torch.cat([p[:, :1] * 3, (p * pl).sum(-1, keepdim=True), rb.nears - rb.fars], dim=-1))
    n = o.shape[0]
    return na.RenderOutput(rgb=rgb, depth=t, weights=torch.zeros(n, 128), s_val=torch.zeros(n, 128),
                           inside_sphere=torch.zeros(n, 128), relax_inside_sphere=torch.zeros(n, 128),
                           analytic_normals=torch.zeros(n, 128, 3), normalized_analytic_normals=torch.zeros(n, 128, 3))


def _bundle(n):
    o, d, pl, near, far = (torch.from_numpy(a) for a in make_rays(n, seed=4))
    return na.RayBundle(origins=o, directions=d, pl_positions=pl, nears=near, fars=far)


def _worker(rank, world, port, n, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        stats = {}
        res = render_sharded(_fake_render, _bundle(n), fields=("rgb", "depth"), stats=stats)
        assert res["rgb"].shape == (n, 3) and res["depth"].shape == (n, 1) and stats["host_s"] > 0.0
        np.save(os.path.join(out_dir, f"rgb_{rank}.npy"), res["rgb"].numpy())
        np.save(os.path.join(out_dir, f"depth_{rank}.npy"), res["depth"].numpy())
        # the same frame with every rank bringing ONLY its slab (no rank holds the other ranks' rays) - and the second frame
        # reuses the first one's gather buffers
        lo, hi = slab_bounds(n, rank, world)
        res2 = render_slab(_fake_render, _bundle(n)[lo:hi], n, fields=("rgb", "depth"))
        assert torch.equal(res2["rgb"], torch.from_numpy(np.load(os.path.join(out_dir, f"rgb_{rank}.npy"))))
        if hi - lo != n:
            with pytest.raises(ValueError):
                render_slab(_fake_render, _bundle(n), n, fields=("rgb",))
    finally:
        dist.destroy_process_group()


def test_slab_bounds_cover_everything():
    for n in (0, 1, 7, 640000, 640001):
        for world in (1, 2, 3, 8):
            b = [slab_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1
    assert views_of_rank(10, 1, 4) == [1, 5, 9] and views_of_rank(10, 0, 2, skip=2) == [0, 4, 8]


@pytest.mark.parametrize("world,n", [(2, 1001), (2, 64), (8, 4096), (8, 4099), (8, 5)])
def test_sharded_render_matches_single_process(tmp_path, world, n):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    ref = _fake_render(_bundle(n))
    for rank in range(world):
        # torch's CPU sigmoid takes different SIMD/remainder paths for different slab lengths: allow 1 ulp
        np.testing.assert_allclose(np.load(tmp_path / f"rgb_{rank}.npy"), ref.rgb.numpy(), rtol=0, atol=2e-7)
        np.testing.assert_allclose(np.load(tmp_path / f"depth_{rank}.npy"), ref.depth.numpy(), rtol=0, atol=1e-6)
    # and the single-process path (no process group) is the plain render
    solo = render_sharded(_fake_render, _bundle(n))
    assert torch.equal(solo["rgb"], ref.rgb)
