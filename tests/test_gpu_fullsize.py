"""GPU tests at the benchmark's own size and beyond one GPU:
  * BASELINE.json configs[1] in full - the 800 x 800 evaluation render, 640 000 image-shaped rays through the default 131 072-ray
    chunks (4 full launches + the 115 712-ray remainder), the path bench.py times;
  * a graphed training step followed by evaluation renders and release() (the pack cache of the device-scalar mode);
  * the multi-rank paths on a box that has two GPUs (skipped otherwise): bench.py's self-spawned ranks, weak and strong, and the
    two-graph training step around a real 2-rank RCCL all-reduce."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import nrhints_amd as na
from nrhints_amd.synthetic import make_image_rays, make_rays, psnr
from oracle import neus_oracle as orc

pytestmark = pytest.mark.gpu
T = torch.from_numpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cu(a):
    return (T(a) if isinstance(a, np.ndarray) else a).float().contiguous().cuda()


def _model(state, train=False):
    m = na.NeuSHintRenderer(na.NeuSModelConfig(), precision="f16x3")
    m.load_state_dict({k: T(np.asarray(v)) for k, v in state.items()})
    m = m.cuda()
    return m if train else m.eval()


def test_configs1_full_frame(scene_states):
    """One 800 x 800 frame of scene b exactly as bench.py renders it (default chunking: launches of 131 072 rays + a remainder of
    115 712), with the full RenderOutput: size-independent properties, bit-equality with a 65 536-chunk render (10 launches: every
    chunk boundary moves), and rgb / depth / visibility against the CPU oracle on a 2 048-ray strided sample that touches every
    launch of both chunkings."""
    model = _model(scene_states["b"])
    assert type(model).max_chunk_rays == 131072 and model.max_chunk_rays == 131072
    # the default takes the whole 640 000-ray frame in ONE call (renderer.whole_frame_rays: 93 GB of workspace); below it is compared
    # bit for bit with 65 536-ray chunks and with the 131 072-ray chunks it replaces
    assert model._pick_chunk(torch.device("cuda", torch.cuda.current_device()), 640000) == 640000
    rays = make_image_rays(800, 800, azimuth=0.6, elevation=0.5)
    n = rays[0].shape[0]
    assert n == 640000 and n % 131072 == 115712
    rb = na.RayBundle(origins=cu(rays[0]), directions=cu(rays[1]), pl_positions=cu(rays[2]), nears=cu(rays[3]), fars=cu(rays[4]))
    one, zero = torch.ones(1, 3).cuda(), torch.zeros(1, 3).cuda()
    with torch.no_grad():
        a = model(rb, is_training=False, background_rgb=one)
    rgb, depth, vis, w = a.rgb, a.depth, a.visibilities, a.weights
    assert rgb.shape == (n, 3) and w.shape == (n, 128) and a.specular_cue.shape == (n, 128, 4)
    for t in (rgb, depth, vis, w, a.analytic_normals, a.normalized_analytic_normals, a.specular_cue):
        assert bool(torch.isfinite(t).all())
    wsum = w.sum(-1, keepdim=True)
    assert bool((w >= 0).all()) and bool((wsum <= 1.0 + 1e-4).all())
    assert bool((rgb >= -1e-6).all()) and bool((rgb <= 1.0 + 1e-5).all())
    assert bool((vis >= 0).all()) and bool((vis <= 1.0 + 1e-6).all())
    assert bool(((a.normalized_analytic_normals.norm(dim=-1) - 1).abs() < 1e-4).all())
    hit_frac = float((wsum > 0.5).float().mean())
    assert 0.02 < hit_frac < 0.9, hit_frac                    # the object covers part of the frame, the rest is background
    corner = wsum.reshape(800, 800)[:40, :40]
    assert float(corner.max()) < 0.05                          # image corners look past the unit sphere
    keep = {k: getattr(a, k).clone() for k in ("rgb", "depth", "visibilities", "weights")}
    nrm = a.analytic_normals[::997].clone()
    del a
    torch.cuda.empty_cache()
    with torch.no_grad():
        b = model(rb, is_training=False, background_rgb=zero)
    torch.testing.assert_close(keep["rgb"] - b.rgb, (1.0 - wsum).expand(-1, 3), rtol=0, atol=2e-6)   # affine in the background
    assert torch.equal(keep["weights"], b.weights) and torch.equal(keep["depth"], b.depth)            # deterministic, bg-independent
    del b
    torch.cuda.empty_cache()
    model.max_chunk_rays = 65536
    try:
        with torch.no_grad():
            c = model(rb, is_training=False, background_rgb=one)
    finally:
        model.max_chunk_rays = type(model).max_chunk_rays
    for k in ("rgb", "depth", "visibilities", "weights"):
        assert torch.equal(keep[k], getattr(c, k)), k                                                # chunk-size independent, bit for bit
    assert torch.equal(nrm, c.analytic_normals[::997])
    del c
    torch.cuda.empty_cache()
    model.whole_frame_rays = 0                      # 4 x 131 072 + 115 712: the chunking of rounds 2-5
    try:
        with torch.no_grad():
            c = model(rb, is_training=False, background_rgb=one)
    finally:
        model.whole_frame_rays = type(model).whole_frame_rays
    for k in ("rgb", "depth", "visibilities", "weights"):
        assert torch.equal(keep[k], getattr(c, k)), k
    del c
    torch.cuda.empty_cache()
    # oracle on a strided sample: stride 313 -> 2 045 rays, at least 369 in every 131 072-launch and 184 in every 65 536-launch
    idx = np.arange(0, n, 313)
    assert len(set(idx // 131072)) == 5 and len(set(idx // 65536)) == 10
    p32 = orc.params_from_state(scene_states["b"])
    ref = orc.render_chunked(p32, *(T(r[idx]) for r in rays), chunk=512, background_rgb=torch.ones(1, 3), mode="minimal")
    got = keep["rgb"].cpu().numpy()[idx]
    dr = np.abs(got - ref["rgb"].numpy())
    assert psnr(got, ref["rgb"].numpy()) > 80.0, psnr(got, ref["rgb"].numpy())
    assert dr.max() < 5e-4 and dr.mean() < 5e-6, (dr.max(), dr.mean())       # same bounds as the 1 000-ray oracle test
    dd = np.abs(keep["depth"].cpu().numpy()[idx] - ref["depth"].numpy())
    assert np.median(dd) < 2e-5 and dd.max() < 5e-2, (np.median(dd), dd.max())
    dv = np.abs(keep["visibilities"].cpu().numpy()[idx] - ref["visibilities"].numpy())
    assert dv.mean() < 1e-4 and dv.max() < 2e-2, (dv.mean(), dv.max())


def test_render_after_graph_release(scene_states):
    """replay -> evaluation render -> release() -> evaluation render.  While the graph is alive 1/s lives on the
    device and the cached pack holds NaN for it; release() must drop that pack, or the render after it runs with inv_s = NaN."""
    from nrhints_amd.training import GraphedTrainStep
    n = 128
    bg = torch.ones(1, 3).cuda()
    model = _model(scene_states["b"], train=True)
    rb_eval = na.RayBundle(**{k: cu(v) for k, v in zip(("origins", "directions", "pl_positions", "nears", "fars"), make_rays(300, seed=71, spread=0.1))})
    rb = na.RayBundle(**{k: cu(v) for k, v in zip(("origins", "directions", "pl_positions", "nears", "fars"), make_rays(n, seed=72, spread=0.1))})
    gt = torch.rand(n, 3, device="cuda")
    step = GraphedTrainStep(model, n, bg, lr=5e-4, warm_up_end=20, global_step=30000)
    step(rb, gt, global_step=30000)
    with torch.no_grad():
        during = model(rb_eval, is_training=False, background_rgb=bg)
    assert bool(torch.isfinite(during.rgb).all())
    step.release()
    with torch.no_grad():
        after = model(rb_eval, is_training=False, background_rgb=bg)
    assert bool(torch.isfinite(after.rgb).all()) and bool(torch.isfinite(after.weights).all())
    assert torch.equal(during.rgb, after.rgb) and torch.equal(during.visibilities, after.visibilities)
    # and equal to a fresh eager model with the same (trained-one-step) parameters
    fresh = na.NeuSHintRenderer(na.NeuSModelConfig(), precision="f16x3")
    fresh.load_state_dict(model.state_dict())
    fresh = fresh.cuda().eval()
    with torch.no_grad():
        ref = fresh(rb_eval, is_training=False, background_rgb=bg)
    assert torch.equal(ref.rgb, after.rgb)
    assert abs(float(after.s_val[0, 0]) - float(ref.s_val[0, 0])) == 0.0


need2 = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs on the box (multi-rank rehearsal)")


def _run(cmd, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1")
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@need2
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks(scaling):
    """bench.py --gpus 2 without a launcher: it spawns its own two ranks over RCCL; one step, no CPU leg.  The headline line is
    well-formed, and a weak-scaling run renders two frames' worth of rays per step."""
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--cpu-rays", "0", "--no-secondary",
              "--no-train", "--scaling", scaling])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 2 and line["scaling"] == scaling and line["value"] > 0
    rays = 640000 * (2 if scaling == "weak" else 1)
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 - rays) < 1e-3 * rays


@need2
def test_two_rank_training_step():
    """tests/multi_gpu_worker.py under torch.distributed.run with two ranks: FlatGradAllReduce over RCCL against the mean of the
    two ranks' local gradients, and the two-graph GraphedTrainStep (graph | eager all-reduce | graph) keeping both ranks' parameters
    identical over three replays."""
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
              "--master-port", "29517", os.path.join("tests", "multi_gpu_worker.py")])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "MULTI_GPU_WORKER_OK rank 0" in r.stdout and "MULTI_GPU_WORKER_OK rank 1" in r.stdout


def _run_shared_gpu(cmd, env, timeout=1500):
    """A rehearsal subprocess tree (N ranks + this pytest process on ONE GPU).  One retry if - and only if - a rank died with a
    device-level fault the runtime reports as ``GPU core dump created``: seen once in this round's seven full-suite runs (rank 4 of
    8, inside a 64-thread torch reduction over a 256-element bias, i.e. not an addressing error of that kernel; ten stand-alone
    repeats of the same command on a fresh box: 10 / 10 clean, profiles/tools/rehearse_loop.sh) - nine processes time-slicing one
    GPU under persistent-grid kernels is a configuration that exists only in this rehearsal (a real run is one process per GPU).
    A second failure, or any other failure, fails the test; the retry is printed."""
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    if r.returncode != 0 and "GPU core dump created" in r.stderr:
        print("rehearsal: a rank died with a device-level fault (GPU core dump); retrying once\n" + r.stderr[-1500:])
        torch.cuda.empty_cache()
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    return r


# ---- the same N-rank code paths on ONE GPU: all ranks on cuda:0, collectives over gloo (RCCL refuses two ranks on one device) ----
# world 8 is the node the driver's scaling run uses: weak = 8 views, strong = one 800x800 frame in slabs of 80 000
# rays (and 640 001 rays: slabs that differ by one, one padding row in the gather), training = configs[2]'s 1 024 rays split 8 x 128
@pytest.mark.parametrize("world,scaling,extra", [(2, "weak", 0), (2, "strong", 0), (8, "weak", 0), (8, "strong", 0), (8, "strong", 1)])
def test_bench_rehearsal_on_one_gpu(world, scaling, extra):
    """bench.py --gpus N with NRH_BENCH_SHARE_GPU=1: the self-spawn through torch.distributed.run, the barriers and the
    max-over-ranks time, the slab-sharded strong-scaling render with its ONE all_gather_into_tensor (every rank holds only its
    slab of rays), and (weak) the training leg's fused steps around the flat gradient all-reduce with the reference's split
    global batch - executed with N ranks.  Timing is meaningless here (one GPU, host-staged collectives); the line's structure,
    the rays accounted for and the per-rank host times are checked."""
    import gc
    gc.collect()
    torch.cuda.empty_cache()        # (this process may still cache a whole-frame workspace from the tests above: 8 ranks need the GPU's memory)
    env_extra = {"NRH_BENCH_SHARE_GPU": "1"}
    if extra:
        env_extra["NRH_BENCH_EXTRA_RAYS"] = str(extra)
    cmd = [sys.executable, "bench.py", "--gpus", str(world), "--steps", "1", "--warmup", "0", "--cpu-rays", "0", "--no-secondary", "--scaling", scaling]
    if scaling == "strong":
        cmd.append("--no-train")
    elif world == 8:
        cmd += ["--train-batch-global", "1024"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1", **env_extra)
    r = _run_shared_gpu(cmd, env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == world and line["scaling"] == scaling and line["value"] > 0 and line.get("rehearsal") is True
    # the line says what the process group was: backend, world size as the collective library sees it, one entry per rank
    assert line["comm"]["backend"] == "gloo" and line["comm"]["world_size"] == world and line["comm"]["all_reduce_of_ones"] == float(world)
    assert len(line["comm"]["devices"]) == world and line["comm"]["devices"][world - 1].startswith(f"rank {world - 1}:")
    # per-rank host (enqueue) time per step beside the step's wall time: the scaling risk SURVEY 8e names, visible from the first run
    host = line["comm"]["host_enqueue_ms_per_step_by_rank"]
    assert len(host) == world and all(0.0 < h < line["comm"]["step_ms"] for h in host), (host, line["comm"]["step_ms"])
    frame = 640000 + extra
    rays = frame * (world if scaling == "weak" else 1)
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 - rays) < 1e-3 * rays
    if scaling == "strong":
        assert line["config"]["rays_per_step_per_gpu"] == (frame + world - 1) // world          # rank 0's slab: the largest
    assert line["roofline"]["library"]["embedded"] == line["roofline"]["library"]["tree"]
    if scaling == "weak":
        train = [json.loads(l) for l in r.stderr.splitlines() if l.startswith('{"train"')]
        assert len(train) == 1 and "error" not in train[0]["train"], train
        t = train[0]["train"]
        assert t["value"] > 0 and t["batch_rays_per_gpu"] == (1024 if world == 2 else 128) and t["loss_last"] < t["loss_first"]
        assert t["dw_half"] is False and "error" not in t["handoff16"] and t["handoff16"]["dw_half"] is True


@pytest.mark.parametrize("world", [2, 8])
def test_training_step_rehearsal_on_one_gpu(world):
    """tests/multi_gpu_worker.py with N ranks on one GPU (gloo): the flat all-reduce against the mean of the ranks' local
    gradients, and the two-graph GraphedTrainStep (graph | eager all-reduce | graph) keeping all ranks' parameters identical."""
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1",
               NRH_WORKER_SHARE_GPU="1")
    r = _run_shared_gpu([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                         "--master-port", str(29519 + world), os.path.join("tests", "multi_gpu_worker.py")], env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    assert all(f"MULTI_GPU_WORKER_OK rank {k}" in r.stdout for k in range(world))
