#!/usr/bin/env python3
"""Record the reference's training step at the batch size BASELINE configs[2] is quoted on (1 024 rays = 131 072 sample
points) and at the three cos-anneal ratios SURVEY.md §8d lists for C3: global_step 0 (ratio 0), 25 000 (0.5), 100 000
(saturated at 1; models/neus_hint_model.py:668-671).  VERDICT r4 "next round" item 1.

Runs ONLY in the build container (imports /root/reference); writes data, never reference source:

    python tests/golden/make_golden_train1024.py          ->  tests/golden/train1024_b.npz
    NRH_GOLDEN_RAYS=128 NRH_GOLDEN_STEPS=0,25000,100000 NRH_GOLDEN_FULL_STEPS=25000 python tests/golden/make_golden_train1024.py
                                                          ->  tests/golden/train128_b.npz
        (the reference's per-rank batch under 8-way DDP with configs[2]'s 1 024 rays: here the 4-wave builds with the 16-bit hand-offs)

Per step s in {0, 25000, 100000} (keys prefixed "s<step>."):
  t_rand_primary / t_rand_shadow   the two torch.rand draws of forward(is_training=True)  (:682, :394)
  loss, rgb_loss, eikonal_loss     float32 run; loss_f64 the float64 run       (pipelines/base_pipeline.py:57-62)
  rgb                              float32 [1024,3] of the float32 run
  grad64.<name>                    d loss / d parameter of the FLOAT64 run, stored as float32 (its rounding, 6e-8 relative,
                                   is four orders below every bound) - 46 tensors; grad64.rays.<origins|directions|pl_positions>
  noise.<name>, noise.rays.<...>   max |grad32 - grad64| over the tensor: the reference's own float32 noise, which is
                                   what tests/conftest.grad_bound turns into the tolerance (full float32 gradients would
                                   double the file for one number per tensor);  noise2.<...>: its L2 norm ||grad32 - grad64||_2
Shared: o, d, pl, near, far (nrhints_amd.synthetic.make_rays(1024, seed=41, spread=0.1)), rgb_gt (seeded uniform colours).
Weights: scene b = perturb_state(scene_a_state.npz) with variance 0.7, as every other *_b fixture.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
from make_golden import REF, _install_stubs  # noqa: E402

STEPS = (0, 25000, 100000)
N = 1024
if os.environ.get("NRH_GOLDEN_RAYS"):          # e.g. NRH_GOLDEN_RAYS=128 NRH_GOLDEN_STEPS=25000 -> train128_b.npz (the 4-wave builds' batch class)
    N = int(os.environ["NRH_GOLDEN_RAYS"])
    STEPS = tuple(int(x) for x in os.environ.get("NRH_GOLDEN_STEPS", "0,25000,100000").split(","))
OUT = "train1024_b.npz" if N == 1024 else f"train{N}_b.npz"
# steps whose gradients / draws are stored in full; the others only leave their per-tensor noise and scale (one number each: what
# the pooled yardstick of tests/test_gpu_train1024.py::_tol needs from them)
FULL = tuple(int(x) for x in os.environ["NRH_GOLDEN_FULL_STEPS"].split(",")) if os.environ.get("NRH_GOLDEN_FULL_STEPS") else STEPS


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)
    import torch

    torch.set_num_threads(8)
    from camera.ray_utils import RayBundle  # reference
    from models.neus_hint_model import NeuSHintRenderer, NeuSModelConfig  # reference

    from nrhints_amd.synthetic import make_rays, perturb_state

    state = perturb_state({k: v for k, v in np.load(os.path.join(HERE, "scene_a_state.npz")).items()})

    def build(dtype):
        torch.manual_seed(0)
        m = NeuSHintRenderer(NeuSModelConfig())
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
        return m.to(dtype).train()

    m32, m64 = build(torch.float32), build(torch.float64)
    o, d, pl, near, far = make_rays(N, seed=41, spread=0.1)
    gt = np.random.RandomState(1024).rand(N, 3).astype(np.float32)
    rec = dict(o=o, d=d, pl=pl, near=near, far=far, rgb_gt=gt, steps=np.asarray(STEPS, np.int64))
    real_rand = torch.rand

    def step(model, dt, gs, rand_fn):
        rays = [torch.from_numpy(a).to(dt).requires_grad_(i < 3) for i, a in enumerate((o, d, pl, near, far))]
        rb = RayBundle(origins=rays[0], directions=rays[1], pl_positions=rays[2], nears=rays[3], fars=rays[4])
        torch.rand = rand_fn
        try:
            r = model(rb, is_training=True, background_rgb=torch.ones(1, 3, dtype=dt), global_step=gs)
        finally:
            torch.rand = real_rand
        g = torch.from_numpy(gt).to(dt)
        # the caller's loss (pipelines/base_pipeline.py:57-62)
        rgb_loss = torch.nn.functional.l1_loss(r.rgb, g, reduction="sum") / (N + 1e-5)
        ge = (torch.linalg.norm(r.analytic_normals, ord=2, dim=-1) - 1.0) ** 2
        eik = (r.relax_inside_sphere * ge).sum() / (r.relax_inside_sphere.sum() + 1e-5)
        loss = rgb_loss + 0.1 * eik
        model.zero_grad()
        loss.backward()
        grads = {n_: p.grad.detach().clone() for n_, p in model.named_parameters()}
        for nm, t in zip(("origins", "directions", "pl_positions"), rays):
            grads["rays." + nm] = t.grad.detach().clone()
        model.zero_grad()
        return r, (loss.detach(), rgb_loss.detach(), eik.detach()), grads

    for gs in STEPS:
        t0 = time.time()
        drawn = []

        def rec_rand(*a, **k):
            t = real_rand(*a, **k)
            drawn.append(t.detach().clone())
            return t

        torch.manual_seed(1000 + gs)
        r32, l32, g32 = step(m32, torch.float32, gs, rec_rand)
        assert len(drawn) == 2, len(drawn)
        replay = [x.double() for x in drawn]
        r64, l64, g64 = step(m64, torch.float64, gs, lambda *a, **k: replay.pop(0))
        assert not replay
        p = f"s{gs}."
        full = gs in FULL
        if full:
            rec[p + "t_rand_primary"], rec[p + "t_rand_shadow"] = drawn[0].numpy(), drawn[1].numpy()
            rec[p + "loss"], rec[p + "rgb_loss"], rec[p + "eikonal_loss"] = (x.numpy() for x in l32)
            rec[p + "loss_f64"], rec[p + "rgb_loss_f64"], rec[p + "eikonal_loss_f64"] = (x.numpy() for x in l64)
            rec[p + "rgb"] = r32.rgb.detach().numpy()
            rec[p + "rgb_f64"] = r64.rgb.detach().numpy().astype(np.float32)
        for k in g64:
            if full:
                rec[p + "grad64." + k] = g64[k].numpy().astype(np.float32)
            rec[p + "gscale." + k] = np.float64(g64[k].abs().max().item())
            rec[p + "noise." + k] = np.float64((g32[k].double() - g64[k]).abs().max().item())
            rec[p + "noise2." + k] = np.float64((g32[k].double() - g64[k]).pow(2).sum().sqrt().item())
        worst = max((float(rec[p + "noise." + k]) / max(float(g64[k].abs().max()), 1e-30), k) for k in g64)
        print(f"step {gs}: loss {float(l32[0]):.6f} (f64 {float(l64[0]):.6f}) eik {float(l32[2]):.5f}; worst f32 noise / scale "
              f"{worst[0]:.2e} on {worst[1]}; {time.time() - t0:.0f} s", flush=True)
    np.savez_compressed(os.path.join(HERE, OUT), **rec)
    print("wrote", OUT, os.path.getsize(os.path.join(HERE, OUT)) / 1e6, "MB")


if __name__ == "__main__":
    main()
