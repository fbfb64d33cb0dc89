#!/usr/bin/env python3
"""Benchmark of the NRHints hot path on MI355X: rendered primary rays / second at 128 samples per ray.

    python bench.py [--gpus N --steps K --warmup W] [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

With ``--gpus N`` (N > 1) and no launcher environment, bench.py starts its own N ranks (one process per GPU through
``torch.distributed.run`` on 127.0.0.1, as the reference's launcher spawns its workers, trainer/launcher.py:41-60).

A *step* is one full 800x800 evaluation render (BASELINE.json configs[1]: 640 000 primary rays, 64 + 64 samples,
one 128-sample shadow ray per primary ray, 4-roughness specular cue) through ``NeuSHintRenderer.forward`` with
the rays already resident in HBM.
  --scaling weak   (default) every rank renders its own view of the scene (rays are independent, no data-path
                   collective: the reference shards evaluation by view too, trainer/trainer.py:288-296)
  --scaling strong ONE 800x800 view split into row blocks over the ranks, pixels all-gathered over RCCL
                   (BASELINE.json configs[3], nrhints_amd/parallel.py: render_sharded)
The scene is the synthetic random-weight scene "b" (reference initialisation under seed 0 + the deterministic
perturbation of nrhints_amd.synthetic.perturb_state, NeuS sharpness 0.7), since datasets/checkpoints are not
reachable offline.

Prints ONE JSON line on rank 0 (contract in the task statement) with these extra objects:
  roofline      dominant kernel (SDF value+feature+gradient at the 128 composite samples), timed live with HIP
                events inside the timed region; algorithmic FLOPs per point are in DESIGN.md section 3
  cpu_baseline  the CPU oracle in "as written" mode (the reference's call pattern, eager fp32 PyTorch) on a bounded
                sample of the same rays: 4096 rays in 512-ray chunks, 1 warm-up + 3 repeats, median (BASELINE.md §4;
                rank 0, N = 1 only)
  secondary     the same render in the other matrix arithmetic (exact fp32 MFMA) - value and roofline fraction
  reduced       the same render in the REDUCED-precision evaluation mode "f16" (the wide SDF kernels in their single-pass builds: one
                fp16 MFMA per K step) - value, roofline fraction, and PSNR against the headline render and the CPU sample; never
                the headline: narrower than the reference's float32 (such a mode must hold PSNR >= 50 dB)
  train         BASELINE.json configs[2]: 1024-ray training steps (forward + backward + Adam; N = 1: the whole step
                replayed as one hipGraph, reported inside the headline line; N > 1: two hipGraphs around one flat RCCL gradient
                all-reduce per step, run AFTER the headline line is out and reported as a second JSON line on stderr;
                --train-batch-global G: the reference's split batch, per-rank = G // N)
  train_small   (N = 1) the graphed step at the reference's per-rank DDP batches, 64 and 128 rays (trainer/trainer.py:116-123)
  train_camopt, register_view   (N = 1) the reference's default preset nr-hints-cam-opt on the fused step; 500-step view registration
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import nrhints_amd as na  # noqa: E402
from nrhints_amd import _lib  # noqa: E402
from nrhints_amd.synthetic import make_image_rays, make_rays, perturb_state, psnr  # noqa: E402

H = W = 800
# algorithmic multiply-accumulates per evaluated point (SURVEY.md §8d / DESIGN.md section 3)
MAC_F_FULL, MAC_F_SDF, MAC_G, MAC_C = 524_544, 459_008, 459_008, 289_792
FLOP_PER_RAY = 2 * (224 * MAC_F_SDF + 128 * (MAC_F_FULL + MAC_G) + 128 * (MAC_F_SDF + MAC_G) + 128 * MAC_C)
FLOP_PER_POINT_CORE = 2 * (MAC_F_FULL + MAC_G)   # the dominant kernel: sdf + feature + gradient per point
# one training ray-step (CHANGELOG.md section 7a): the evaluation path without the reflectance forward's share of the no-grad pass,
# plus training forward (F + C), tangent and value sweeps (2 G + F), reflectance adjoint (C) and the weight gradients
FLOP_PER_RAY_STEP = 1.4186e9
# MI355X_MICROARCH.md dense MFMA peaks: fp32-input 157.3 TFLOP/s; fp16 2 500 TFLOP/s.  The f16x3 mode spends three
# fp16 MFMAs per algorithmic multiply-add, so its ceiling in ALGORITHMIC flops is 833 TFLOP/s; frac is quoted
# against the fp16 peak all the same (the honest denominator for the instruction that is issued).
PEAK_TFLOPS = {"f32": 157.3, "f16x3": 2500.0, "f16": 2500.0}
# measured context of the f16x3 roofline fraction (reported beside it, never instead of it): v_mfma_f32_32x32x16_f16 with random
# operands in a bare register loop on one MI355X under its 1.4 kW package limit (2 492 TFLOP/s with all-zero operands), and the
# kernel's MFMA-FLOP per algorithmic FLOP from the PMC passes (profiles/r06/mfma_sustained.log, profiles/r06/pmc_f16x3_*/summary.txt)
SUSTAINED_MFMA_TFLOPS = 1710.0
MFMA_FLOP_PER_ALGORITHMIC_FLOP = 3.16
# the CPU baseline is the reference's algorithm; what runs on the GPU box is its restatement (oracle, mode "as_written").
# Their wall times on identical rays / threads, reference under no_grad as its evaluation loop calls it
# (pipelines/base_pipeline.py:114-119): 7.47 s against 7.42 s per 512 rays on 8 cores (profiles/r05/cpu_baseline_crosscheck.log).
REF_OVER_PORT_TIME = 1.006


def build_scene(precision):
    torch.manual_seed(0)
    model = na.NeuSHintRenderer(na.NeuSModelConfig(), precision=precision)
    state_a = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
    state_b = perturb_state(state_a)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state_b.items()})
    if os.environ.get("NRH_BENCH_NO_FUSE"):       # A/B of NrhNet.feat_fused (profiles/r02/fused_head_ab.log); not a product knob
        model.fuse_feature_head = False
    if os.environ.get("NRH_BENCH_CHUNK"):          # A/B of the rays-per-launch chunk (profiles/r02/chunk_rays_ab.log, r06/chunk_ab.log)
        model.max_chunk_rays = int(os.environ["NRH_BENCH_CHUNK"])
        model.whole_frame_rays = 0
    if os.environ.get("NRH_BENCH_SHARE_GPU") == "1" or os.environ.get("NRH_BENCH_WHOLE_FRAME") == "0":
        model.whole_frame_rays = 0            # N ranks rehearsing on ONE GPU must not each take a whole-frame workspace; A/B runs
    if os.environ.get("NRH_BENCH_SHADOW_JVP"):     # A/B of the forward-mode shadow evaluation (profiles/r02/shadow_jvp_ab.log: slower)
        model.shadow_jvp = True
    if os.environ.get("NRH_BENCH_NO_WIDE_COLOR"):  # A/B of the wide reflectance kernel (profiles/r02/wide_color_ab.log)
        model.wide_color = False
    return model, state_b


def dominant_kernel(precision, wide):
    if precision == "f16" and wide:
        return "nrh32t::sdf32_kernel<2>", "sdf32_kernel<2>, one-term build (single-pass f16: sdf + feature + d sdf/dx, 128 pts/ray)"
    if precision == "f16x3" and wide:
        return "nrh32::sdf32_kernel<2>", "sdf32_kernel<2> (wide f16x3: sdf + feature + d sdf/dx, 128 pts/ray)"
    return "sdf_kernel<2, %s>" % {"f16x3": "1", "f16": "1", "f32": "0"}[precision], "sdf_kernel<2> (sdf + feature + d sdf/dx, 128 pts/ray)"


def kernel_source_hash():
    """Hash of the sources the evaluation kernels are generated / compiled from: a committed counter summary is only quoted when
    it was taken with the same kernels (profiles/pmc_run.sh records this value in its summary)."""
    import hashlib
    h = hashlib.sha256()
    for name in ("gen_mlp32.py", "nrh_mlp32.h", "nrh_sdf32.hip", "nrh_color32.hip", "nrh_wide.hip", "nrh_mlp.h", "nrh_sdf.hip", "nrh_common.h"):
        with open(os.path.join(ROOT, "nrhints_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _traffic_from_summary(txt, precision, wide):
    import re
    m = re.search(re.escape(dominant_kernel(precision, wide)[0]) + r"[^\n]*\n((?:   .*\n)+)", txt)
    if not m:
        return None
    vals = dict(re.findall(r"(\w+)\s+n=\s*\d+ mean=([0-9.e+]+)", m.group(1)))
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        # FETCH_SIZE [KiB] x 2 (gfx950 counts wide coalesced reads at half their size, MI355X_MICROARCH.md) + WRITE_SIZE [KiB]
        return int((2.0 * float(vals["FETCH_SIZE"]) + float(vals["WRITE_SIZE"])) * 1024)
    return None


def pmc_traffic(precision, wide, rays_per_launch=None):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 --pmc passes of this same command
    (profiles/pmc_run.sh) - IF that summary was taken with today's kernel sources (its ``source_hash`` line); a summary of other
    kernels is reported as stale and its number is not quoted.  -> (bytes or None, path or None, kind).
    ``rays_per_launch``: this run's average rays per launch of the kernel; the summary's counters are per launch of ITS run
    (``rays_per_launch`` line; summaries older than round 6 were taken with 4 x 131 072 + 115 712 rays per frame = 128 000 on
    average) and the kernel's traffic is proportional to its points (the sigma' scratch round trip per point dominates), so the
    figure is scaled to this run's launch size."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"pmc_{precision}_v*", "summary.txt")))
    for path in reversed(files):
        txt = open(path).read()
        t = _traffic_from_summary(txt, precision, wide)
        if t is None:
            continue
        m = re.search(r"source_hash\s+(\w+)", txt)
        rel = os.path.relpath(path, ROOT)
        mr = re.search(r"rays_per_launch\s+([0-9.]+)", txt)
        if rays_per_launch:
            t = int(t * float(rays_per_launch) / (float(mr.group(1)) if mr else 128000.0))
        if m and m.group(1) == kernel_source_hash():
            return t, rel, "committed rocprofv3 --pmc summary of this command taken with the same kernel sources (source_hash matches); --pmc collects it live"
        return None, rel, ("stale: the newest committed counter summary was taken with other kernel sources (source_hash "
                           f"{m.group(1) if m else 'absent'} != {kernel_source_hash()}); run bench.py --pmc or profiles/pmc_run.sh")
    return None, None, "no committed counter summary for this precision; run bench.py --pmc"


def pmc_live(precision, wide):
    """--pmc: collect FETCH_SIZE and WRITE_SIZE of one frame live - two rocprofv3 passes (one counter each: they do not fit one
    pass; --kernel-trace only, as the pool requires) of this script with --steps 1 --no-train --no-secondary --cpu-rays 0."""
    import glob
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not found"
    out = tempfile.mkdtemp(prefix="nrh_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", NRH_BENCH_PMC_CHILD="1")
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", os.path.join(out, ctr), "--", sys.executable,
               os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--cpu-rays", "0", "--no-train", "--no-secondary", "--precision", precision]
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600, check=False)
        except Exception as e:  # noqa: BLE001
            return None, f"rocprofv3 failed: {e}"
    key = dominant_kernel(precision, wide)[0]
    import csv
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        v = []
        for f in glob.glob(os.path.join(out, ctr, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                v += [float(r["Counter_Value"]) for r in csv.DictReader(fh) if key in r["Kernel_Name"] and r["Counter_Name"] == ctr]
        if not v:
            return None, f"no {ctr} rows for {key}"
        vals[ctr] = sum(v) / len(v)
    shutil.rmtree(out, ignore_errors=True)
    return int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024), "live: two rocprofv3 --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE) of one frame of this command"


def cpu_baseline(state, rays_np, n_sample, gpu_rgb, budget_s=150.0):
    """Time the oracle ("as written": 13 SDF forwards + autograd gradient per render) on the host, BASELINE.md §4 protocol:
    ``n_sample`` rays (4096) in 512-ray chunks, 1 warm-up + 3 timed repeats, median.

    Eager PyTorch on [512*128, 256] operands stops scaling long before a 256-core host is full (oversubscribed it
    is 10x slower), so the thread count is calibrated on ONE 512-ray chunk - the size the timed repeats use - over
    {16, 32, 64, 128, all} cores (stopping once past the optimum) and the best one is used and reported as ``cores`` (the chunk
    doubles as the warm-up).  The
    repeats stop early if the time budget runs out (said in ``sample``)."""
    from oracle import neus_oracle as orc  # the checker; only this leg and the tests import it
    host = os.cpu_count() or 1
    p = orc.params_from_state(state)
    idx = np.linspace(0, rays_np[0].shape[0] - 1, n_sample).astype(np.int64)
    sub = [torch.from_numpy(a[idx]) for a in rays_np]
    bg = torch.ones(1, 3)
    best, best_t = host, float("inf")
    calib = {}
    # ascending, and no further once a count is 1.5x slower than the best so far: past the optimum (16 - 32 threads on the 256-core
    # boxes) the time only grows - 4.8 s at 64, 10.7 s at 128, 133 s at 256 threads (profiles/r04/bench_v2.json) - and measuring
    # that tail again in every run cost 2.5 minutes of the bench's wall time
    for th in sorted({min(host, 16), min(host, 32), min(host, 64), min(host, 128), host}):
        torch.set_num_threads(th)
        t0 = time.perf_counter()
        orc.render_chunked(p, *(t[:512] for t in sub), chunk=512, background_rgb=bg, mode="as_written")
        dt = time.perf_counter() - t0
        calib[th] = round(dt, 2)
        if dt < best_t:
            best, best_t = th, dt
        elif dt > 1.5 * best_t:
            break
    torch.set_num_threads(best)
    t_start = time.perf_counter()
    times, out = [], None
    for rep in range(3):
        t0 = time.perf_counter()
        out = orc.render_chunked(p, *sub, chunk=512, background_rgb=bg, mode="as_written")
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start + times[-1] > budget_s:
            break
    med = float(np.median(times))
    ref = out["rgb"].numpy()
    return {"value": round(n_sample / med, 2), "unit": "rays/s", "cores": best, "host_cores": host, "kind": "port",
            # time of the imported reference / time of this port on the same 512 rays and threads, measured where the reference
            # exists (the build container; profiles/cpu_baseline_crosscheck.py -> profiles/r05/cpu_baseline_crosscheck.log)
            "ref_over_port_time": REF_OVER_PORT_TIME, "ref_over_port_source": "profiles/r05/cpu_baseline_crosscheck.log",
            "sample": f"{n_sample} rays strided over the benchmark frame in 512-ray chunks, oracle mode=as_written (reference "
                      f"call pattern), fp32 PyTorch eager, one-chunk warm-up + {len(times)} repeat(s), median {med:.1f} s",
            "repeats_s": [round(t, 2) for t in times], "thread_calibration_s_per_512_rays": calib,
            "psnr_gpu_vs_cpu_db": round(psnr(gpu_rgb[idx], ref), 2),
            "max_abs_rgb_diff": float(np.abs(gpu_rgb[idx] - ref).max())}


def timed_render(model, rb, bg, steps, warmup, dist, dev, sharded, n_total=None):
    """W untimed + K timed render steps bracketed by barrier + synchronize; -> (seconds, last output, kernel ms, launches, host
    seconds of THIS rank inside the timed steps: wall time its Python spent enqueueing, nothing in a step synchronises the device).
    ``sharded``: ``rb`` is this rank's SLAB of a frame of ``n_total`` rays (parallel.render_slab: one all_gather_into_tensor)."""
    from nrhints_amd import parallel
    stats = {}

    def step():
        with torch.no_grad():
            if sharded:
                return parallel.render_slab(lambda r: model(r, is_training=False, background_rgb=bg), rb, n_total, fields=("rgb",), stats=stats)
            t0 = time.perf_counter()
            out = model(rb, is_training=False, background_rgb=bg)
            stats["host_s"] = stats.get("host_s", 0.0) + time.perf_counter() - t0
            return out

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    out = None
    for _ in range(warmup):
        out = step()
    lib = _lib.load()
    _lib.check(lib.nrh_kernel_timing_select(2), "timing_select")
    fence()
    stats.clear()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    k_ms, k_n = ctypes.c_double(0), ctypes.c_longlong(0)
    _lib.check(lib.nrh_kernel_timing_read(ctypes.byref(k_ms), ctypes.byref(k_n)), "timing_read")
    _lib.check(lib.nrh_kernel_timing_select(-1), "timing_select")
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, out, k_ms.value, max(1, k_n.value), float(stats.get("host_s", 0.0))


def orbit_views(ncam=12, radius=3.6, elevation=0.4):
    """Poses [ncam,4,4] of cameras on an orbit looking at the origin and one light per view (the synthetic stand-in for a real
    scene's calibrated views: BASELINE configs[3] / [4] name datasets that are not reachable offline)."""
    poses = np.zeros((ncam, 4, 4), dtype=np.float32)
    for i, a in enumerate(np.linspace(0.2, 5.9, ncam)):
        pos = radius * np.array([np.cos(elevation) * np.cos(a), np.cos(elevation) * np.sin(a), np.sin(elevation)])
        fwd = -pos / np.linalg.norm(pos)
        right = np.cross(fwd, [0.0, 0.0, 1.0]); right /= np.linalg.norm(right)
        poses[i, :3, :3] = np.stack([right, np.cross(right, fwd), -fwd], axis=1)
        poses[i, :3, 3], poses[i, 3, 3] = pos, 1.0
    pls = (poses[:, :3, 3] * 1.2 + np.array([0.3, -0.2, 0.5])).astype(np.float32)
    return poses, pls


def camopt_legs(dev, batch=1024, steps=30, warm=3, view_steps=500, view_batch=512):
    """The reference's DEFAULT preset, nr-hints-cam-opt (configs/main_config.py:60-64: RayGeneratorConfig(cam_opt_mode="SO3xR3")):
      train_camopt   1024-pixel training steps that start at the data loader's RawPixelBundle - ray generation with per-view pose
                     deltas, the fused step incl. the ray adjoints, the ray generator's adjoint, Adam over both parameter groups
                     (trainer/trainer.py:99-102) - replayed as one hipGraph
      register_view  what evaluation does per view under that preset (pipelines/base_pipeline.py:71-91, :103-106): 500 Adam steps
                     of 512 random pixels on the view's pose delta with the renderer frozen; seconds per view"""
    from nrhints_amd import RawPixelBundle, RayGenerator, RayGeneratorConfig
    from nrhints_amd.pipeline import CameraModel
    from nrhints_amd.training import GraphedTrainStep, register_view
    torch.manual_seed(0)
    ncam, Hc, Wc = 12, 200, 200
    cam = CameraModel(H=Hc, W=Wc, cx=Wc / 2, cy=Hc / 2, fx=280.0, fy=280.0)
    poses, pls = orbit_views(ncam)
    student = na.NeuSHintRenderer(na.NeuSModelConfig()).to(dev)
    teacher, _ = build_scene(student.precision)
    teacher = teacher.to(dev).eval()
    bg = torch.ones(1, 3, device=dev)
    rg_true = RayGenerator(cam, ncam, RayGeneratorConfig()).to(dev)
    rs = np.random.RandomState(7)
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def pixels(n, seed_rs, view=None):
        img = seed_rs.randint(0, ncam, size=n) if view is None else np.full(n, view)
        kw = dict(img_indices=cu(img[:, None].astype(np.int64)), h_indices=cu(seed_rs.randint(0, Hc, size=(n, 1)).astype(np.float32)),
                  w_indices=cu(seed_rs.randint(0, Wc, size=(n, 1)).astype(np.float32)), poses=cu(poses[img]), pls=cu(pls[img]))
        with torch.no_grad():      # ground-truth pixels are data: rendered before the timed region
            gt = teacher(rg_true(RawPixelBundle(rgb_gt=None, **kw)), background_rgb=bg).rgb
        return RawPixelBundle(rgb_gt=gt, **kw)

    out = {}
    # ---- training under cam-opt ----
    rg = RayGenerator(cam, ncam, RayGeneratorConfig(cam_opt_mode="SO3xR3")).to(dev)
    batches = [pixels(batch, rs) for _ in range(steps + warm)]
    step = GraphedTrainStep(student, batch, bg, warm_up_end=10, global_step=20000, ray_generator=rg, ray_lr=rg.config.opt_lr)
    losses, t0, block_s = [], None, []
    for blk in range(3):                   # three consecutive blocks of `steps` steps, the median one is the value (train_leg: repeats)
        for i, pb in enumerate(batches):
            if blk and i < warm:
                continue
            if i == warm:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            losses.append(step(pb, pb.rgb_gt, global_step=20000 + blk * len(batches) + i)["loss"])
        torch.cuda.synchronize()
        block_s.append(time.perf_counter() - t0)
    dt = sorted(block_s)[1]
    fused = bool(step._use_fused)
    step.release()
    out["train_camopt"] = {"metric": "training ray-steps/s under nr-hints-cam-opt (ray generation with pose deltas + forward + backward + ray "
                                     "adjoints + Adam over both groups)", "value": round(batch * steps / dt, 1), "unit": "ray-steps/s",
                           "batch_rays": batch, "steps": steps, "warmup": warm, "ms_per_step": round(dt / steps * 1e3, 3),
                           "repeats_ms": [round(b / steps * 1e3, 3) for b in block_s], "views": ncam,
                           "mode": "hipGraph replay of the " + ("fused (autograd-free) step" if fused else "autograd path"),
                           "loss_first": round(float(losses[0]), 5), "loss_last": round(float(losses[-1]), 5),
                           "pose_delta_moved": float(rg.cam_pose_adjustment.detach().abs().max())}
    # ---- register_view: one view, its pose perturbed by a known shift, 500 steps ----
    model = teacher
    rg2 = RayGenerator(cam, ncam, RayGeneratorConfig(cam_opt_mode="SO3xR3")).to(dev)
    hh, ww = np.meshgrid(np.arange(Hc, dtype=np.float32), np.arange(Wc, dtype=np.float32), indexing="ij")
    view = 3
    with torch.no_grad():
        full_true = RawPixelBundle(img_indices=torch.full((Hc * Wc, 1), view, dtype=torch.long, device=dev), h_indices=cu(hh.reshape(-1, 1)),
                                   w_indices=cu(ww.reshape(-1, 1)), poses=cu(poses[view]).expand(Hc * Wc, 4, 4).contiguous(),
                                   pls=cu(pls[view]).expand(Hc * Wc, 3).contiguous(), rgb_gt=None)
        gt = model(rg_true(full_true), background_rgb=bg).rgb.reshape(Hc, Wc, 3).cpu()
    shifted = poses[view].copy()
    shifted[:3, 3] += np.array([0.05, -0.03, 0.04], dtype=np.float32)
    img = RawPixelBundle(img_indices=torch.full((Hc, Wc, 1), view, dtype=torch.long), h_indices=torch.from_numpy(hh)[..., None],
                         w_indices=torch.from_numpy(ww)[..., None], poses=torch.from_numpy(shifted).expand(Hc, Wc, 4, 4),
                         pls=torch.from_numpy(pls[view]).expand(Hc, Wc, 3), rgb_gt=gt)
    gen = torch.Generator().manual_seed(11)
    register_view(model, rg2, img, dev, steps=20, batch_size=view_batch, lr=1e-3, generator=gen)      # warm-up: caches, pack plans
    with torch.no_grad():
        rg2.cam_pose_adjustment.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr = register_view(model, rg2, img, dev, steps=view_steps, batch_size=view_batch, lr=1e-3, generator=gen)
    torch.cuda.synchronize()
    dtv = time.perf_counter() - t0
    out["register_view"] = {"metric": "seconds per registered view (500 pose-refinement steps of 512 pixels, renderer frozen)",
                            "value": round(dtv, 3), "unit": "s/view", "higher_is_better": False, "steps": view_steps, "batch_rays": view_batch,
                            "ms_per_step": round(dtv / view_steps * 1e3, 3), "loss_first10": round(float(np.mean(tr[:10])), 5),
                            "loss_last10": round(float(np.mean(tr[-10:])), 5),
                            "note": "excluded from rays/s (SURVEY 8d, config C4); the reference also computes and discards all 46 parameter "
                                    "gradients in these steps - skipped here, results identical"}
    return out


HANDOFF_LABEL = {False: "f16x3 (float32 dW hand-offs: three-term products throughout - the precision-matched form)",
                 True: "f16x3 + fp16 dW hand-offs (weight-gradient products: ONE fp16 MFMA pass on operands rounded to 11 bits, range-scaled; "
                       "narrower than the reference's float32 in those products - labelled option NeuSHintRenderer.dw_half)"}


def train_leg(dev, rank, world, dist, batch=1024, steps=30, warm=3, global_batch=None, repeats=1, dw_half=False):
    """BASELINE.json configs[2] (and [4] for N > 1): 1024-ray training steps of the reference-initialised student against
    pixels of scene b (rendered before the timed region: ground-truth pixels are data).  ``global_batch``: the reference's DDP
    semantics - the batch is split, per-rank = global // world (trainer/trainer.py:116-123) - instead of a fixed per-rank batch.
    ``repeats`` > 1 (the millisecond-scale small-batch legs, whose 40 steps are a 40 ms sample): that many consecutive blocks of
    ``steps`` steps are timed, every block's ms per step is reported (``repeats_ms``) and the MEDIAN block is the value - inside the
    full bench run one block of a leg was seen 25 % slow (1.22 against 0.97 ms at 64 rays, profiles/r05/bench_v4.json against
    bench_legs_half_ab.log of the same tree) where the leg alone, or after the 1 024-ray leg only, was not."""
    from nrhints_amd.training import FlatGradAllReduce, GraphedTrainStep
    torch.manual_seed(0)
    if global_batch:
        batch = max(16, global_batch // world)
    student = na.NeuSHintRenderer(na.NeuSModelConfig()).to(dev)
    student.dw_half = bool(dw_half)          # numerics option of the fused step (renderer attribute, never the environment)
    teacher, _ = build_scene(student.precision)
    teacher = teacher.to(dev).eval()
    bg = torch.ones(1, 3, device=dev)
    batches = []
    nblocks = max(1, int(repeats)) if world == 1 else 1
    for s in range(steps + warm):
        o, d, pl, near, far = (torch.from_numpy(a).to(dev) for a in make_rays(batch, seed=1000 + 97 * rank + s, spread=0.08))
        rb = na.RayBundle(origins=o, directions=d, pl_positions=pl, nears=near, fars=far)
        with torch.no_grad():
            batches.append((rb, teacher(rb, background_rgb=bg).rgb))
    # one hipGraph per step at one rank; with more ranks two graphs around ONE flat RCCL all-reduce of the gradients
    # (training.GraphedTrainStep: forward + loss + backward + flattening | all-reduce | unflatten + Adam)
    sync = None
    if world > 1:
        sync = FlatGradAllReduce(list(student.parameters()))
        sync.broadcast_parameters(0)
    # NRH_BENCH_COLLECTIVE_IN_GRAPH=1 (opt-in, RCCL only): the all-reduce captured inside the step's one hipGraph
    in_graph = bool(sync is not None and os.environ.get("NRH_BENCH_COLLECTIVE_IN_GRAPH") == "1" and not SHARE_GPU)
    graphed = GraphedTrainStep(student, batch, bg, warm_up_end=10, global_step=20000, grad_sync=sync, collective_in_graph=in_graph)
    run = lambda i, rb, gt: graphed(rb, gt, global_step=20000 + i)
    losses = []
    t0 = None
    block_s = []
    for blk in range(nblocks):
        for i, (rb, gt) in enumerate(batches):
            if blk and i < warm:
                continue                       # (later blocks replay the timed batches only)
            if i == warm:
                if dist is not None:
                    dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            losses.append(run(blk * len(batches) + i, rb, gt)["loss"])
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        block_s.append(time.perf_counter() - t0)
    dt = sorted(block_s)[len(block_s) // 2]
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    graphed.release()
    losses = [float(x) for x in losses]
    value = world * batch * steps / dt
    peak = PEAK_TFLOPS[student.precision]
    return {"metric": "training ray-steps/s (forward + backward + Adam)", "value": round(value, 1), "unit": "ray-steps/s",
            "batch_rays_per_gpu": batch, "steps": steps, "warmup": warm, "ms_per_step": round(dt / steps * 1e3, 3),
            "dtype": HANDOFF_LABEL[bool(dw_half)] if student.precision == "f16x3" else student.precision, "dw_half": bool(dw_half),
            "global_batch": batch * world,
            "mode": "hipGraph replay" if world == 1 else ("one hipGraph incl. the flat RCCL all-reduce" if in_graph else
                                                            "two hipGraphs around one flat RCCL all-reduce per step"),
            "loss_first": round(float(losses[0]), 5), "loss_last": round(float(losses[-1]), 5),
            **({"repeats_ms": [round(b / steps * 1e3, 3) for b in block_s]} if nblocks > 1 else {}),
            "bound_note": "the step's big kernels (SDF training forward, dW, tangent / value sweeps, reflectance adjoint) move the saved "
                          "activations through HBM at 4-5.5 TB/s (profiles/r05/pmc_train_summary.txt, DESIGN.md section 3); the MFMA "
                          "fraction below is the SURVEY 8d convention",
            "roofline": {"bound": "mfma", "algorithmic_gflop_per_ray_step": round(FLOP_PER_RAY_STEP / 1e9, 4),
                         "achieved": round(value * FLOP_PER_RAY_STEP / 1e12 / world, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(value * FLOP_PER_RAY_STEP / 1e12 / world / peak, 4)}}


# test-only: NRH_BENCH_SHARE_GPU=1 runs the N ranks of --gpus N on ONE GPU with gloo collectives, so that the self-spawn, the
# barriers, the strong-scaling render and the gradient exchange of the training leg execute with N > 1 on a one-GPU box
# (tests/test_gpu_fullsize.py); the line it prints carries "rehearsal": true
SHARE_GPU = os.environ.get("NRH_BENCH_SHARE_GPU") == "1"


def respawn(args):
    """--gpus N without a launcher: start N ranks ourselves (one process per GPU) and relay their output."""
    if torch.cuda.device_count() < args.gpus and not SHARE_GPU:
        raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) are visible")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--cpu-rays", type=int, default=4096, help="rays in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--precision", choices=sorted(PEAK_TFLOPS), default=na.NeuSHintRenderer.precision)
    ap.add_argument("--no-train", action="store_true", help="skip the training leg (configs[2])")
    ap.add_argument("--no-secondary", action="store_true", help="skip the render in the other precision")
    ap.add_argument("--no-camopt", action="store_true", help="skip the nr-hints-cam-opt legs (training under pose refinement, register_view)")
    ap.add_argument("--train-batch-global", type=int, default=0,
                    help="training leg with the reference's DDP semantics: this GLOBAL batch split over the ranks (per-rank = global // N, "
                         "trainer/trainer.py:116-123) instead of 1024 rays per rank")
    ap.add_argument("--pmc", action="store_true", help="collect roofline.traffic live (two rocprofv3 --pmc passes of one frame; +1-2 min)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP hot path has no CPU fallback)")
    if SHARE_GPU:
        local_rank = 0          # rehearsal of the N-rank code paths on a box with ONE GPU: every rank on cuda:0, collectives over gloo
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        # RCCL on ROCm ("nccl"): barrier, max-over-ranks, pixel all-gather, gradient all-reduce.  RCCL refuses two ranks on one device,
        # hence gloo (which stages device tensors through the host) for the one-GPU rehearsal - its numbers are not a benchmark
        dist.init_process_group(backend="gloo" if SHARE_GPU else "nccl")
    # what the process group actually is (reported in the line, so that a multi-GPU run certifies itself): backend, world size as
    # the collective library sees it, and every rank's device
    comm = {"backend": "none", "world_size": 1, "devices": [f"cuda:{local_rank} {torch.cuda.get_device_name(local_rank)}"]}
    if dist is not None:
        try:
            props = torch.cuda.get_device_properties(local_rank)
            mine = f"rank {rank}: cuda:{local_rank} {torch.cuda.get_device_name(local_rank)} pci {getattr(props, 'pci_bus_id', '?')} pid {os.getpid()}"
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe)                       # a device collective through the backend: sum of ones = ranks that took part
            comm = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "all_reduce_of_ones": float(probe.item()),
                    "devices": gathered}
        except Exception as e:  # noqa: BLE001 - a report, never a reason for the benchmark to stop
            comm = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "error": f"{type(e).__name__}: {e}"[:200]}

    model, state = build_scene(args.precision)
    peak = PEAK_TFLOPS[args.precision]
    model = model.to(dev).eval()
    strong = args.scaling == "strong"
    # weak: each rank renders its own view of the same scene (different azimuth / light); strong: everyone holds view 0
    rays_np = make_image_rays(H, W, azimuth=0.6 + (0.0 if strong else 0.7 * rank), elevation=0.5)
    nrays = H * W + int(os.environ.get("NRH_BENCH_EXTRA_RAYS", "0"))      # (test hook: a frame that does not divide over the ranks)
    if nrays != H * W:
        rays_np = tuple(np.concatenate([a, a[: nrays - H * W]], axis=0) for a in rays_np)
    sharded = strong and world > 1
    # strong scaling: a rank uploads and holds ONLY its slab of the frame (parallel.slab_bounds: contiguous row blocks)
    from nrhints_amd.parallel import slab_bounds
    lo, hi = slab_bounds(nrays, rank, world) if sharded else (0, nrays)
    rb = na.RayBundle(**{k: torch.from_numpy(np.ascontiguousarray(v[lo:hi])).to(dev) for k, v in
                         zip(("origins", "directions", "pl_positions", "nears", "fars"), rays_np)})
    bg = torch.ones(1, 3, device=dev)
    dt, out, k_ms, launches, host_s = timed_render(model, rb, bg, args.steps, args.warmup, dist, dev, sharded=sharded, n_total=nrays)
    rgb = (out["rgb"] if isinstance(out, dict) else out.rgb).cpu().numpy()
    # per-rank host time outside kernels (the >= 6x risk at 8 GPUs is host overhead, not bandwidth): every rank's
    # enqueue time per step beside the step's wall time
    host_ms = round(host_s / max(1, args.steps) * 1e3, 3)
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, host_ms)
        comm["host_enqueue_ms_per_step_by_rank"] = per_rank
    else:
        comm["host_enqueue_ms_per_step_by_rank"] = [host_ms]
    comm["step_ms"] = round(dt / max(1, args.steps) * 1e3, 3)

    secondary = None
    if rank == 0 and world == 1 and not args.no_secondary:
        other = "f32" if args.precision == "f16x3" else "f16x3"
        m2, _ = build_scene(other)
        m2 = m2.to(dev).eval()
        sec_steps = 3                              # (one frame in exact fp32 takes ~4 s; VERDICT r4: more than a single sample)
        dt2, out2, k2_ms, l2, _ = timed_render(m2, rb, bg, sec_steps, 1, None, dev, sharded=False)
        ach2 = FLOP_PER_POINT_CORE * (nrays * 128 * sec_steps / l2) / (k2_ms / l2 * 1e-3) / 1e12
        secondary = {"dtype": other, "value": round(nrays * sec_steps / dt2, 1), "unit": "rays/s", "steps": sec_steps, "warmup": 1,
                     "roofline_achieved_tflops": round(ach2, 2), "roofline_frac": round(ach2 / PEAK_TFLOPS[other], 4),
                     "psnr_vs_primary_db": round(psnr(out2.rgb.cpu().numpy(), rgb), 2)}
        del m2, out2
        torch.cuda.empty_cache()
    reduced = None
    if rank == 0 and world == 1 and not args.no_secondary and args.precision == "f16x3":
        try:
            m3, _ = build_scene("f16")
            m3 = m3.to(dev).eval()
            red_steps = 3
            dt3, out3, k3_ms, l3, _ = timed_render(m3, rb, bg, red_steps, 1, None, dev, sharded=False)
            ach3 = FLOP_PER_POINT_CORE * (nrays * 128 * red_steps / l3) / (k3_ms / l3 * 1e-3) / 1e12
            rgb3 = out3.rgb.cpu().numpy()
            reduced = {"dtype": "f16 (one fp16 MFMA pass in the SDF kernels; reflectance net and per-ray stages as f16x3)",
                       "value": round(nrays * red_steps / dt3, 1), "unit": "rays/s", "steps": red_steps, "warmup": 1,
                       "ms_per_step": round(dt3 / red_steps * 1e3, 3), "avg_launch_ms": round(k3_ms / l3, 3),
                       "roofline_achieved_tflops": round(ach3, 2), "roofline_frac": round(ach3 / PEAK_TFLOPS["f16"], 4),
                       "psnr_vs_primary_db": round(psnr(rgb3, rgb), 2), "max_abs_rgb_vs_primary": float(np.abs(rgb3 - rgb).max()),
                       "note": "reduced precision: not the headline (narrower than the reference's float32); gate for such a mode: PSNR >= 50 dB"}
            del m3, out3
            torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001
            reduced = {"error": f"{type(e).__name__}: {e}"[:300]}

    def run_train_leg():
        """configs[2]'s step twice: ``train`` itself at float32 hand-offs (the precision-matched number; the default of the
        library), and beside it, labelled, the same step with the fp16 dW hand-offs (``train["handoff16"]``)."""
        try:
            leg = train_leg(dev, rank, world, dist, global_batch=args.train_batch_global or None, dw_half=False)
        except Exception as e:  # the headline must survive a failure of the secondary leg
            return {"error": f"{type(e).__name__}: {e}"[:300]}
        try:
            h = train_leg(dev, rank, world, dist, global_batch=args.train_batch_global or None, dw_half=True)
            leg["handoff16"] = {k: h[k] for k in ("value", "unit", "ms_per_step", "dtype", "dw_half", "loss_first", "loss_last")}
            leg["handoff16"]["roofline_frac"] = h["roofline"]["frac"]
        except Exception as e:  # noqa: BLE001
            leg["handoff16"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        return leg

    # One rank: the training leg runs first and rides in the headline line.  Several ranks: a failure on ONE rank inside the
    # leg (OOM, RCCL error) would leave the others blocked in its gradient all-reduce, so the headline line is printed BEFORE
    # the leg starts and the leg's result follows as a second JSON line on stderr ({"train": ...}); nothing after the
    # headline can keep it from being printed.
    train = run_train_leg() if (not args.no_train and world == 1) else None
    train_small = None
    if rank == 0 and world == 1 and not args.no_train and not args.no_camopt:
        # the reference splits the global batch over the ranks (trainer/trainer.py:116-123: 512 / 8 = 64 rays per rank, 128 with
        # configs[2]'s 1 024): what ONE such per-rank step costs here, as the graph replay a rank runs
        try:
            legs = {b: train_leg(dev, rank, 1, None, batch=b, steps=40, repeats=3) for b in (64, 128)}
            train_small = {"metric": "graphed training step at the reference's per-rank DDP batch (one GPU, no exchange); median of 3 blocks "
                                     "of 40 steps", "unit": "ms per step",
                           "ms_per_step": {str(b): l["ms_per_step"] for b, l in legs.items()},
                           "repeats_ms": {str(b): l["repeats_ms"] for b, l in legs.items()},
                           "ray_steps_per_s": {str(b): l["value"] for b, l in legs.items()}, "steps": 40, "dtype": legs[64]["dtype"]}
            # ... and the 128-ray step with the fp16 hand-offs, labelled (64 rays runs on the channel-split kernels, which have no such form)
            h128 = train_leg(dev, rank, 1, None, batch=128, steps=40, repeats=3, dw_half=True)
            train_small["handoff16_128"] = {"ms_per_step": h128["ms_per_step"], "repeats_ms": h128["repeats_ms"], "dtype": h128["dtype"]}
        except Exception as e:  # noqa: BLE001
            train_small = {"error": f"{type(e).__name__}: {e}"[:300]}
    camopt = None
    if rank == 0 and world == 1 and not args.no_train and not args.no_camopt:
        try:
            camopt = camopt_legs(dev)
        except Exception as e:  # noqa: BLE001
            camopt = {"error": f"{type(e).__name__}: {e}"[:300]}

    if rank == 0:
        work = 1 if strong else world                       # strong: the job is ONE frame however many ranks render it
        value = work * nrays * args.steps / dt
        pts_per_launch = (nrays / (world if strong else 1)) * 128 * args.steps / launches
        avg_ms = k_ms / launches
        achieved = FLOP_PER_POINT_CORE * pts_per_launch / (avg_ms * 1e-3) / 1e12
        wide = bool(getattr(model, "wide_kernels", False)) and args.precision == "f16x3"
        traffic, traffic_src, traffic_kind = pmc_traffic(args.precision, wide, pts_per_launch / 128.0)
        if args.pmc and world == 1 and not os.environ.get("NRH_BENCH_PMC_CHILD"):
            live, how = pmc_live(args.precision, wide)
            if live is not None:
                traffic, traffic_src, traffic_kind = live, None, how
            else:
                traffic_kind += f" (live collection failed: {how})"
        line = {
            "metric": "rendered rays/sec (128 samples/ray)", "value": round(value, 1), "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "800x800 eval render (640000 primary rays/step" + ("" if strong else "/GPU") + "), 64+64 samples/ray, "
                                   "shadow + specular hints, synthetic random-weight scene b (BASELINE configs[1])",
                       "rays_per_step_per_gpu": (hi - lo), "samples_per_ray": 128,
                       "chunk_rays": int(model._pick_chunk(dev, nrays // (world if strong else 1))),
                       "parallelism": (f"one view in {world} row blocks + RCCL all-gather of rgb" if strong else f"view-sharded x{world}"),
                       "algorithmic_gflop_per_ray": round(FLOP_PER_RAY / 1e9, 4),
                       "whole_path_tflops": round(value * FLOP_PER_RAY / 1e12, 2),
                       "whole_path_frac_of_mfma_peak": round(value * FLOP_PER_RAY / 1e12 / world / peak, 4)},
            "roofline": {"bound": "mfma", "kernel": dominant_kernel(args.precision, wide)[1],
                         "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_unit": "bytes/launch (PMC)",
                         "traffic_source": traffic_src,
                         "traffic_kind": traffic_kind, "kernel_source_hash": kernel_source_hash(),
                         # the loaded binary's embedded source hash against the tree's (nrhints_amd/build_id.py): _lib.load() has
                         # already refused a stale default library; a variant named through NRHINTS_HIP_LIB shows up here
                         "library": _lib.library_identity(),
                         # bytes the kernel's contract moves (features + sdf + gradient out, 1 044 B per point) and SURVEY 8(d)'s
                         # algorithmic figure for the whole path (44 B in + 6 676 B out = 6 720 B per ray)
                         "algorithmic_bytes_per_launch": int(pts_per_launch * 1044),
                         "survey_8d_bytes_per_launch": int(pts_per_launch / 128 * 6720),
                         "avg_launch_ms": round(avg_ms, 3), "launches": int(launches),
                         "algorithmic_flop_per_point": FLOP_PER_POINT_CORE},
        }
        if args.precision == "f16x3":
            # Context for `frac` (which stays priced against the guide's 2 500 TFLOP/s): the same MFMA instruction in a bare register
            # loop sustains SUSTAINED_MFMA_TFLOPS with random operands - the package power limit, not the issue rate (32.00 cycles per
            # MFMA there), sets it (profiles/ubench/mfma_sustained.hip -> profiles/r06/mfma_sustained.log); the kernel issues
            # MFMA_FLOP_PER_ALGORITHMIC_FLOP MFMA-FLOP per algorithmic FLOP (three-term split, bias MFMAs, K padding; PMC SQ_INSTS_MFMA)
            line["roofline"]["power_limit_context"] = {
                "sustained_mfma_tflops_random_operands": SUSTAINED_MFMA_TFLOPS, "source": "profiles/r06/mfma_sustained.log",
                "mfma_flop_per_algorithmic_flop": MFMA_FLOP_PER_ALGORITHMIC_FLOP,
                "issued_mfma_tflops": round(achieved * MFMA_FLOP_PER_ALGORITHMIC_FLOP, 1),
                "frac_of_sustained": round(achieved * MFMA_FLOP_PER_ALGORITHMIC_FLOP / SUSTAINED_MFMA_TFLOPS, 3)}
        if world == 1 and args.cpu_rays > 0:
            line["cpu_baseline"] = cpu_baseline(state, rays_np, args.cpu_rays, rgb)
        else:
            line["cpu_baseline"] = None
        line["comm"] = comm
        if SHARE_GPU:
            line["rehearsal"] = True
        line["secondary"] = secondary
        if reduced is not None:
            if isinstance(line.get("cpu_baseline"), dict) and "error" not in reduced:
                reduced["cpu_sample_note"] = "PSNR against the CPU oracle sample is reported for the headline render (cpu_baseline.psnr_gpu_vs_cpu_db)"
            line["reduced"] = reduced
        line["train"] = train if world == 1 or args.no_train else "second JSON line on stderr (multi-rank run)"
        if train_small is not None:
            line["train_small"] = train_small
        if camopt is not None:
            line.update(camopt)
        print(json.dumps(line), flush=True)
    if world > 1 and not args.no_train:
        train = run_train_leg()
        if rank == 0:
            print(json.dumps({"train": train}), file=sys.stderr, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
