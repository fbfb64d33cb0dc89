#!/usr/bin/env python3
"""Is nrh_render_forward_train a pure function of its inputs?  Three calls on the 1 024-ray fixture batch (fresh output tensors each
time, the workspace reused, a fused training step in between): every exported array must be bit-identical."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nrhints_amd as na  # noqa: E402
from nrhints_amd import train_fused  # noqa: E402
from nrhints_amd.synthetic import perturb_state  # noqa: E402

T = torch.from_numpy


def main():
    gs = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "train1024_b.npz")))
    st = perturb_state(dict(np.load(os.path.join(ROOT, "tests", "golden", "scene_a_state.npz"))))
    p = f"s{gs}."
    cu = lambda a: T(np.asarray(a)).float().contiguous().cuda()
    model = na.NeuSHintRenderer(na.NeuSModelConfig(), precision=prec)
    model.load_state_dict({k: T(np.asarray(v)) for k, v in st.items()})
    model = model.cuda()
    o, d, pl, near, far = (cu(g[k]) for k in ("o", "d", "pl", "near", "far"))
    tp, ts = cu(g[p + "t_rand_primary"]).reshape(-1).contiguous(), cu(g[p + "t_rand_shadow"]).contiguous()
    ca = min(1.0, gs / 50000)

    def call():
        r = model._render_train(o, d, pl, near.reshape(-1), far.reshape(-1), ca, tp, ts, 0)
        torch.cuda.synchronize()
        return {k: r[k].clone() for k in ("mid_z", "dists", "weights", "normals", "visibilities", "cue", "depth", "inside")} | \
            {"sdf": r["pre"]["sdf"].clone(), "feat": r["pre"]["feat"].clone()}

    a = call()
    torch.empty(1 << 28, device="cuda").fill_(float("nan"))       # dirty 1 GiB of freed memory between the calls
    torch.cuda.synchronize()
    b = call()
    rb = na.RayBundle(origins=o, directions=d, pl_positions=pl, nears=near, fars=far)
    train_fused.train_step_backward(model, rb, cu(g["rgb_gt"]), torch.ones(1, 3).cuda(), gs, t_rand_primary=tp, t_rand_shadow=ts)
    torch.cuda.synchronize()
    c = call()
    for name, x, y in (("call 1 vs call 2 (memory dirtied in between)", a, b), ("call 1 vs call 3 (after a fused step)", a, c)):
        bad = {k: (int((x[k] != y[k]).sum()), float((x[k] - y[k]).abs().max())) for k in x if not torch.equal(x[k], y[k])}
        print(name, "->", "bit-identical" if not bad else f"DIFFERENT: {bad}")


if __name__ == "__main__":
    main()
