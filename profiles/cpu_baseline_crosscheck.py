#!/usr/bin/env python3
"""Is bench.py's `cpu_baseline` (the oracle in mode "as_written") time-faithful to the reference?  SURVEY.md section 8d: the CPU
restatement "must agree within noise" with the imported reference's wall time on identical rays and threads.

BUILD CONTAINER ONLY (imports /root/reference, which does not exist on the GPU box): times, on the same 512 rays of the benchmark
frame (the reference's inference_chunk_size) and the same thread count,
  reference   NeuSHintRenderer.forward(is_training=False) under no_grad, as pipelines/base_pipeline.py:114-119 calls it
  port        oracle.neus_oracle.render_forward(mode="as_written") - what bench.py times on the GPU box's host cores
alternating, 1 warm-up + REPEATS timed runs each, and prints one JSON line with the medians and `ref_over_port_time`.
bench.py carries that factor in cpu_baseline (REF_OVER_PORT_TIME) when the two differ by more than noise.

    python profiles/cpu_baseline_crosscheck.py [threads] [repeats]  >  profiles/r05/cpu_baseline_crosscheck.log
"""
import json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_golden import REF, _install_stubs  # noqa: E402


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    _install_stubs()
    sys.path.insert(0, REF)
    sys.path.insert(0, ROOT)
    import torch
    torch.set_num_threads(threads)
    from camera.ray_utils import RayBundle  # reference
    from models.neus_hint_model import NeuSHintRenderer, NeuSModelConfig  # reference
    from nrhints_amd.synthetic import make_image_rays, perturb_state
    from oracle import neus_oracle as orc

    state = perturb_state({k: v for k, v in np.load(os.path.join(ROOT, "tests", "golden", "scene_a_state.npz")).items()})
    torch.manual_seed(0)
    ref = NeuSHintRenderer(NeuSModelConfig())
    ref.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    ref.eval()
    rays = make_image_rays(800, 800, azimuth=0.6, elevation=0.5)
    idx = np.linspace(0, rays[0].shape[0] - 1, 4096).astype(np.int64)[:512]          # bench.py's first 512-ray chunk
    sub = [torch.from_numpy(a[idx]) for a in rays]
    bg = torch.ones(1, 3)
    p = orc.params_from_state(state)

    def run_ref():
        rb = RayBundle(origins=sub[0], directions=sub[1], pl_positions=sub[2], nears=sub[3], fars=sub[4])
        with torch.no_grad():
            return ref(rb, is_training=False, background_rgb=bg).rgb

    def run_port():
        return orc.render_forward(p, *sub, background_rgb=bg, mode="as_written")["rgb"]

    t_ref, t_port = [], []
    a, b = run_ref(), run_port()              # warm-up of both
    for _ in range(repeats):
        t0 = time.perf_counter(); a = run_ref(); t_ref.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); b = run_port(); t_port.append(time.perf_counter() - t0)
    mr, mp = float(np.median(t_ref)), float(np.median(t_port))
    print(json.dumps({"rays": 512, "threads": threads, "torch": torch.__version__, "reference_s": [round(t, 2) for t in t_ref],
                      "port_as_written_s": [round(t, 2) for t in t_port], "reference_median_s": round(mr, 2), "port_median_s": round(mp, 2),
                      "ref_over_port_time": round(mr / mp, 3), "reference_rays_per_s": round(512 / mr, 1), "port_rays_per_s": round(512 / mp, 1),
                      "max_abs_rgb_diff": float((a - b).abs().max())}))


if __name__ == "__main__":
    main()
