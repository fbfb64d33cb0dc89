#!/usr/bin/env python3
"""Benchmark of the NRHints hot path on MI355X: rendered primary rays / second at 128 samples per ray.

    python bench.py [--gpus N --steps K --warmup W]                 (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A *step* is one full 800x800 evaluation render (BASELINE.json configs[1]: 640 000 primary rays, 64 + 64 samples,
one 128-sample shadow ray per primary ray, 4-roughness specular cue) through ``NeuSHintRenderer.forward`` with
the rays already resident in HBM.  With N > 1 every rank renders its own view of the scene (rays are independent,
no data-path collective: the reference shards evaluation by view too, trainer/trainer.py:288-296) - weak scaling.
The scene is the synthetic random-weight scene "b" (reference initialisation under seed 0 + the deterministic
perturbation of nrhints_amd.synthetic.perturb_state, NeuS sharpness 0.7), since datasets/checkpoints are not
reachable offline.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      dominant kernel (SDF value+feature+gradient at the 128 composite samples), timed live with HIP
                events inside the timed region; algorithmic FLOPs per point are in DESIGN.md
  cpu_baseline  the CPU oracle in "as written" mode (the reference's call pattern, eager fp32 PyTorch) on a bounded
                sample of the same rays, all host cores (rank 0, N = 1 only)
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import nrhints_amd as na  # noqa: E402
from nrhints_amd import _lib  # noqa: E402
from nrhints_amd.synthetic import make_image_rays, perturb_state, psnr  # noqa: E402

H = W = 800
# algorithmic multiply-accumulates per evaluated point (SURVEY.md §8d / DESIGN.md §4)
MAC_F_FULL, MAC_F_SDF, MAC_G, MAC_C = 524_544, 459_008, 459_008, 289_792
FLOP_PER_RAY = 2 * (224 * MAC_F_SDF + 128 * (MAC_F_FULL + MAC_G) + 128 * (MAC_F_SDF + MAC_G) + 128 * MAC_C)
FLOP_PER_POINT_CORE = 2 * (MAC_F_FULL + MAC_G)   # the dominant kernel: sdf + feature + gradient per point
# MI355X_MICROARCH.md dense MFMA peaks: fp32-input 157.3 TFLOP/s; fp16 2 500 TFLOP/s.  The f16x3 mode spends three
# fp16 MFMAs per algorithmic multiply-add, so its ceiling in ALGORITHMIC flops is 833 TFLOP/s; frac is quoted
# against the fp16 peak all the same (the honest denominator for the instruction that is issued).
PEAK_TFLOPS = {"f32": 157.3, "f16x3": 2500.0}


def build_scene(precision):
    torch.manual_seed(0)
    model = na.NeuSHintRenderer(na.NeuSModelConfig(), precision=precision)
    state_a = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
    state_b = perturb_state(state_a)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state_b.items()})
    return model, state_b


def pmc_traffic(precision):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
    (profiles/pmc_run.sh): FETCH_SIZE [KiB] x 2 (gfx950 counts wide coalesced reads at half their size) + WRITE_SIZE
    [KiB].  None if no counter summary for this precision is committed."""
    import glob
    import re
    tag = {"f16x3": "1", "f32": "0"}[precision]
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"pmc_{precision}_v*", "summary.txt")))
    if not files:
        return None, None
    txt = open(files[-1]).read()
    m = re.search(r"sdf_kernel<2, %s>\n((?:   .*\n)+)" % tag, txt)
    if not m:
        return None, None
    vals = dict(re.findall(r"(\w+)\s+n=\s*\d+ mean=([0-9.e+]+)", m.group(1)))
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None, None
    return int((2.0 * float(vals["FETCH_SIZE"]) + float(vals["WRITE_SIZE"])) * 1024), os.path.relpath(files[-1], ROOT)


def cpu_baseline(state, rays_np, n_sample, gpu_rgb):
    """Time the oracle ("as written": 13 SDF forwards + autograd gradient per render, 512-ray chunks) on the host.

    Eager PyTorch on [512*128, 256] operands stops scaling long before a 256-core host is full (oversubscribed it
    is 10x slower), so the thread count is calibrated on a 64-ray probe over {all, 64, 32, 16} cores and the best
    one is used and reported as ``cores``."""
    from oracle import neus_oracle as orc  # the checker; only this leg and the tests import it
    host = os.cpu_count() or 1
    p = orc.params_from_state(state)
    idx = np.linspace(0, rays_np[0].shape[0] - 1, n_sample).astype(np.int64)
    sub = [torch.from_numpy(a[idx]) for a in rays_np]
    bg = torch.ones(1, 3)
    best, best_t = host, float("inf")
    for th in sorted({host, min(host, 64), min(host, 32), min(host, 16)}, reverse=True):
        torch.set_num_threads(th)
        t0 = time.perf_counter()
        orc.render_chunked(p, *(t[:64] for t in sub), chunk=512, background_rgb=bg, mode="as_written")
        dt = time.perf_counter() - t0
        if dt < best_t:
            best, best_t = th, dt
        if dt > 20.0:
            continue
    torch.set_num_threads(best)
    t0 = time.perf_counter()
    out = orc.render_chunked(p, *sub, chunk=512, background_rgb=bg, mode="as_written")
    dt = time.perf_counter() - t0
    ref = out["rgb"].numpy()
    return {"value": round(n_sample / dt, 2), "unit": "rays/s", "cores": best, "host_cores": host, "kind": "port",
            "sample": f"{n_sample} rays strided over the benchmark frame, one 512-ray chunk shape, oracle "
                      f"mode=as_written (reference call pattern), fp32 PyTorch eager, {dt:.1f} s",
            "psnr_gpu_vs_cpu_db": round(psnr(gpu_rgb[idx], ref), 2),
            "max_abs_rgb_diff": float(np.abs(gpu_rgb[idx] - ref).max())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-rays", type=int, default=512, help="rays in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--precision", choices=sorted(PEAK_TFLOPS), default=na.NeuSHintRenderer.precision)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP hot path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl")  # RCCL on ROCm; used for the barrier + max-over-ranks only

    model, state = build_scene(args.precision)
    peak = PEAK_TFLOPS[args.precision]
    model = model.to(dev).eval()
    # each rank renders its own view of the same scene (different azimuth / light), rays resident in HBM
    rays_np = make_image_rays(H, W, azimuth=0.6 + 0.7 * rank, elevation=0.5)
    rb = na.RayBundle(**{k: torch.from_numpy(v).to(dev) for k, v in
                         zip(("origins", "directions", "pl_positions", "nears", "fars"), rays_np)})
    bg = torch.ones(1, 3, device=dev)
    nrays = H * W

    def step():
        with torch.no_grad():
            return model(rb, is_training=False, background_rgb=bg)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    lib = _lib.load()
    _lib.check(lib.nrh_kernel_timing_select(2), "timing_select")
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
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

    if rank == 0:
        value = world * nrays * args.steps / dt
        launches = max(1, k_n.value)
        pts_per_launch = nrays * 128 * args.steps / launches
        avg_ms = k_ms.value / launches
        achieved = FLOP_PER_POINT_CORE * pts_per_launch / (avg_ms * 1e-3) / 1e12
        traffic, traffic_src = pmc_traffic(args.precision)
        line = {
            "metric": "rendered rays/sec (128 samples/ray)", "value": round(value, 1), "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "800x800 eval render (640000 primary rays/step/GPU), 64+64 samples/ray, "
                                   "shadow + specular hints, synthetic random-weight scene b (BASELINE configs[1])",
                       "rays_per_step_per_gpu": nrays, "samples_per_ray": 128,
                       "chunk_rays": int(model.max_chunk_rays), "parallelism": f"view-sharded x{world}",
                       "algorithmic_gflop_per_ray": round(FLOP_PER_RAY / 1e9, 4),
                       "whole_path_tflops": round(value * FLOP_PER_RAY / 1e12, 2),
                       "whole_path_frac_of_mfma_peak": round(value * FLOP_PER_RAY / 1e12 / world / peak, 4)},
            "roofline": {"bound": "mfma", "kernel": "sdf_kernel<2> (sdf + feature + d sdf/dx, 128 pts/ray)",
                         "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_unit": "bytes/launch (PMC)",
                         "traffic_source": traffic_src, "algorithmic_bytes_per_launch": int(pts_per_launch * 1044),
                         "avg_launch_ms": round(avg_ms, 3), "launches": int(launches),
                         "algorithmic_flop_per_point": FLOP_PER_POINT_CORE},
        }
        if world == 1 and args.cpu_rays > 0:
            line["cpu_baseline"] = cpu_baseline(state, rays_np, args.cpu_rays, out.rgb.cpu().numpy())
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
