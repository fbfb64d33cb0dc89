"""Training step with the outside-NeRF background: fused step (background island between the kernels) against forward() + backward(),
eager and captured; ms per step at 512 and 1 024 rays.  python profiles/outside_step_bench.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import nrhints_amd as na
from nrhints_amd import train_fused, training
from nrhints_amd.synthetic import make_rays, perturb_state


def main():
    a = dict(np.load("tests/golden/scene_a_state.npz"))
    cfg = na.NeuSModelConfig(renderer=na.NeuSRendererConfig(use_outside_nerf=True))
    bg = torch.ones(1, 3).cuda()
    for n in (512, 1024):
        rays = make_rays(n, seed=5, spread=0.2)
        rb = na.RayBundle(**{k: torch.from_numpy(v).cuda() for k, v in zip(("origins", "directions", "pl_positions", "nears", "fars"), rays)})
        gt = torch.full((n, 3), 0.5).cuda()
        res = {}
        for mode in ("autograd", "fused", "captured"):
            torch.manual_seed(0)
            m = na.NeuSHintRenderer(cfg)
            sd = m.state_dict()
            sd.update({k: torch.from_numpy(np.asarray(v)) for k, v in perturb_state(a).items()})
            m.load_state_dict(sd)
            m = m.cuda().train()
            opt = training.make_optimizer(m)
            if mode == "captured":
                step = training.GraphedTrainStep(m, n, bg, global_step=20000)
                run = lambda it: step(rb, gt, global_step=20000 + it)
            else:
                o, sch = opt if isinstance(opt, tuple) else (opt, None)
                run = lambda it: training.train_step(m, rb, gt, bg, 20000 + it, o, fused=(mode == "fused"))
            for it in range(5):
                run(it)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            K = 20
            for it in range(K):
                run(5 + it)
            torch.cuda.synchronize()
            res[mode] = (time.perf_counter() - t0) / K * 1e3
            if mode == "captured":
                step.release()
        print(f"{n} rays, use_outside_nerf: " + ", ".join(f"{k} {v:.2f} ms" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
