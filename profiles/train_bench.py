#!/usr/bin/env python3
"""Training-step throughput (BASELINE config 3: 1024-ray batches, forward + backward + Adam) on one GPU.

Synthetic task: fit the reference-initialised model (scene a) to pixels rendered from scene b.  Prints ray-steps/s and
the loss trajectory (the teacher renders the ground-truth pixels before the timed region).  The no-grad stages run in the HIP kernels, the differentiable render_core in autograd_core."""
import os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nrhints_amd as na
from nrhints_amd.synthetic import make_rays, perturb_state
from nrhints_amd.training import make_optimizer, train_step

def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    torch.manual_seed(0)
    student = na.NeuSHintRenderer().cuda()
    student.dw_half = "half" in sys.argv[4:]
    if "nofuse" in sys.argv[4:]:                   # A/B of the fused "SDF pass + sampler step" launch (nrh_sampler_fusion: measurements only)
        from nrhints_amd import _lib as _l
        _l.load().nrh_sampler_fusion(0)
    if "forcefuse" in sys.argv[4:]:
        from nrhints_amd import _lib as _l
        _l.load().nrh_sampler_fusion(2)       # 4th argument "half": the fp16 dW hand-offs (NeuSHintRenderer.dw_half); default float32
    backend = sys.argv[3] if len(sys.argv) > 3 else "hip"          # hip | hip_nosync (no per-step loss read-back) | graph | manual | autograd (the last two: tests/torch_backends.py)
    teacher = na.NeuSHintRenderer()
    st = perturb_state({k: v.detach().cpu().numpy().copy() for k, v in student.state_dict().items()})
    teacher.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
    teacher = teacher.cuda().eval()
    bg = torch.ones(1, 3, device="cuda")
    opt, sched = make_optimizer(student, warm_up_end=10)
    # ground-truth pixels are data in a real run: render them with the teacher BEFORE the timed region
    batches = []
    for step in range(steps + 3):
        o, d, pl, near, far = (torch.from_numpy(a).cuda() for a in make_rays(batch, seed=1000 + step, spread=0.08))
        rb = na.RayBundle(origins=o, directions=d, pl_positions=pl, nears=near, fars=far)
        with torch.no_grad():
            batches.append((rb, teacher(rb, background_rgb=bg).rgb))
    losses, t0 = [], None
    graphed = None
    if len(sys.argv) > 3 and sys.argv[3] == "graph":       # the whole step captured into one hipGraph
        from nrhints_amd.training import GraphedTrainStep
        graphed = GraphedTrainStep(student, batch, bg, warm_up_end=10, global_step=20000)
    import contextlib
    ctx = contextlib.nullcontext()
    if backend in ("manual", "autograd"):
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from tests.torch_backends import use_torch_backend
        ctx = use_torch_backend(student, backend)
    ctx.__enter__()
    for step, (rb, gt) in enumerate(batches):
        if step == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        if graphed is not None:
            out = graphed(rb, gt, global_step=20000 + step)
        else:
            out = train_step(student, rb, gt, bg, global_step=20000 + step, optimizer=opt, scheduler=sched, sync=backend != "hip_nosync")
        losses.append(out["loss"])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    losses = [float(x) for x in losses]
    print(json.dumps({"metric": "training ray-steps/s (fwd+bwd+Adam)", "batch": batch, "steps": steps,
                      "value": round(batch * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 2),
                      "loss_first3": [round(x, 5) for x in losses[:3]], "loss_last3": [round(x, 5) for x in losses[-3:]],
                      "precision": student.precision, "dw_half": bool(student.dw_half), "sdf_backward": backend, "hip_graph": graphed is not None}))

if __name__ == "__main__":
    main()
