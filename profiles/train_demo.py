#!/usr/bin/env python3
"""End-to-end check that the HIP training path LEARNS: a student (reference initialisation) is fitted to pixels rendered
from a teacher scene (synthetic scene b: perturbed geometry, sharper variance) with the reference's loss / Adam / schedule,
and PSNR is tracked on a held-out 128x128 view.  Prints one JSON line per evaluation and a final summary line.

    python profiles/train_demo.py [steps=1500] [batch=1024] [late_eval_every=0] [seed=0] [student=default|narrow]

student = narrow: the student is a NARROWER network than the kernels are compiled for (sdf 128 / multi_res 4 / feature 128,
reflectance 128 / multi_res 2: zero-padded onto the compiled kernels, packing.pad_to_compiled) fitted to the same full-size teacher.

late_eval_every > 0: from step steps - 500 on, the held-out view is also evaluated every that many steps, and the summary carries the
MEAN of those PSNRs and of the last 200 training losses - single evaluations of this fit swing by 3-4 dB from one to the next (in
every run of every round), which is too coarse to compare two arithmetic variants by.
"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nrhints_amd as na
from nrhints_amd.synthetic import make_image_rays, make_rays, perturb_state, psnr
from nrhints_amd.training import make_optimizer, train_step


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    late = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 0            # another stream of training rays and jitter (same scene, same view)
    late_psnr, losses = [], []
    narrow = len(sys.argv) > 5 and sys.argv[5] == "narrow"
    torch.manual_seed(0)
    teacher = na.NeuSHintRenderer()
    st = perturb_state({k: v.detach().cpu().numpy().copy() for k, v in teacher.state_dict().items()})
    torch.manual_seed(0)
    student = na.NeuSHintRenderer(na.NeuSModelConfig(sdf_network=na.SDFNetConfig(d_hidden=128, multi_res=4, d_out_feat=128),
                                                     reflectance_network=na.ReflectanceNetConfig(d_hidden=128, multi_res=2))
                                  if narrow else None).cuda()
    teacher.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
    teacher = teacher.cuda().eval()
    bg = torch.ones(1, 3, device="cuda")
    opt, sched = make_optimizer(student, warm_up_end=100)
    bundle = lambda rays: na.RayBundle(**{k: torch.from_numpy(v).cuda() for k, v in
                                           zip(("origins", "directions", "pl_positions", "nears", "fars"), rays)})
    eval_rb = bundle(make_image_rays(128, 128, focal=178.0, azimuth=0.3, elevation=0.4))
    with torch.no_grad():
        eval_gt = teacher(eval_rb, background_rgb=bg).rgb.cpu().numpy()

    def evaluate(step, t_train):
        student.eval()
        with torch.no_grad():
            rgb = student(eval_rb, background_rgb=bg).rgb.cpu().numpy()
        student.train()
        line = {"step": step, "eval_psnr_db": round(psnr(rgb, eval_gt), 2), "train_seconds": round(t_train, 2)}
        print(json.dumps(line), flush=True)
        return line["eval_psnr_db"]

    first = evaluate(0, 0.0)
    torch.manual_seed(seed)
    t_train, last_loss = 0.0, None
    for step in range(steps):
        rb = bundle(make_rays(batch, seed=5000 + step + 100000 * seed, spread=0.08))
        with torch.no_grad():
            gt = teacher(rb, background_rgb=bg).rgb           # "dataset" pixels
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = train_step(student, rb, gt, bg, global_step=60000 + step, optimizer=opt, scheduler=sched)
        torch.cuda.synchronize(); t_train += time.perf_counter() - t0
        last_loss = out["loss"]
        losses.append(float(last_loss))
        if (step + 1) % 250 == 0:
            last = evaluate(step + 1, t_train)
            if late and step + 1 > steps - 500:
                late_psnr.append(last)
        elif late and step + 1 > steps - 500 and (step + 1) % late == 0:
            late_psnr.append(evaluate(step + 1, t_train))
    print(json.dumps({"summary": "student fitted to teacher pixels through the HIP training kernels", "steps": steps, "batch": batch,
                      "eval_psnr_first_db": first, "eval_psnr_last_db": last, "final_loss": round(last_loss, 5),
                      "ray_steps_per_s": round(steps * batch / t_train, 1), "precision": student.precision,
                      "sdf_backward": "hip", "seed": seed,
                      **({"late_eval_psnr_mean_db": round(float(np.mean(late_psnr)), 2), "late_eval_psnr_min_max_db": [min(late_psnr), max(late_psnr)],
                          "late_evals": len(late_psnr), "mean_loss_last_200": round(float(np.mean(losses[-200:])), 5)} if late_psnr else {})}), flush=True)


if __name__ == "__main__":
    main()
