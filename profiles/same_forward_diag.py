#!/usr/bin/env python3
"""Round 6 diagnostic for tests/test_gpu_train1024.py: WHERE does the HIP training forward's rgb differ from the float64 oracle
when placement (sections, visibility, cue) AND the SDF network's outputs (sdf, gradient, feature) are shared?  Per ray: rgb, the
per-sample weights and the per-sample colours of both sides; the worst rays in detail.

    python profiles/same_forward_diag.py [global_step] [precision]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nrhints_amd as na  # noqa: E402
from nrhints_amd import train_fused  # noqa: E402
from nrhints_amd.synthetic import perturb_state  # noqa: E402
from oracle import neus_oracle as orc  # noqa: E402   (diagnostic script: the checker, not the product)
from tests.placement import hip_placement  # noqa: E402

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
    rays_c = tuple(cu(g[k]) for k in ("o", "d", "pl", "near", "far"))
    rb = na.RayBundle(origins=rays_c[0], directions=rays_c[1], pl_positions=rays_c[2], nears=rays_c[3], fars=rays_c[4])
    fwd = {}
    train_fused.train_step_backward(model, rb, cu(g["rgb_gt"]), torch.ones(1, 3).cuda(), gs, t_rand_primary=cu(g[p + "t_rand_primary"]),
                                    t_rand_shadow=cu(g[p + "t_rand_shadow"]), forward_out=fwd)
    z, vis, cue, net, (mid, dist) = hip_placement(fwd)          # the step's OWN forward (a separate call may place samples differently)
    B = next(iter(model._fused_buffers.values()))
    n = 1024
    rgb_hip = B.rgb.cpu().double().numpy()
    col_hip = B.color.cpu().double().numpy().reshape(n, 128, 3)
    w_hip = fwd["weights"].cpu().double().numpy()
    cos_anneal = min(1.0, gs / 50000)
    res = dict(inside=fwd["inside"])
    params = orc.params_from_state({k: T(np.asarray(v)).double() for k, v in st.items()}, torch.float64)
    r64 = [T(np.asarray(g[k])).double() for k in ("o", "d", "pl", "near", "far")]
    for label, netov in (("same placement only", None), ("same placement + same sdf / grad / feat", net)):
        with torch.no_grad():
            out = orc.render_forward(params, *r64, background_rgb=torch.ones(1, 3, dtype=torch.float64), is_training=True, global_step=gs,
                                     t_rand_primary=T(g[p + "t_rand_primary"]).double(), t_rand_shadow=T(g[p + "t_rand_shadow"]).double(),
                                     mode="minimal", z_override=z, vis_override=vis, cue_override=cue, net_override=netov,
                                     sections_override=(mid, dist), keep_intermediates=True)
        rgb = out["rgb"].numpy()
        w = out["weights"].numpy()
        col = out["sampled_color"].numpy()
        drgb = np.abs(rgb_hip - rgb).max(1)
        dw = np.abs(w_hip - w)
        dc = np.abs(col_hip - col).max(2)
        print(f"== {label}: max |rgb| diff {drgb.max():.3e} (median ray {np.median(drgb):.2e}); max |w| diff {dw.max():.3e}; "
              f"max |colour| diff {dc.max():.3e}; max |colour| diff weighted by w {np.max(dc * w):.3e}")
        worst = np.argsort(-drgb)[:4]
        if netov is not None and len(sys.argv) > 3:        # dump the worst rays' per-sample arrays for offline analysis
            rr = worst
            np.savez(sys.argv[3], rays=rr, sdf_hip=net["sdf"].numpy().reshape(n, 128)[rr], grad_hip=net["grad"].numpy().reshape(n, 128, 3)[rr],
                     mid=mid.numpy()[rr], dist=dist.numpy()[rr], w_hip=w_hip[rr], w_orc=w[rr], alpha_orc=out["alpha"].numpy()[rr],
                     sdf_orc=sdf[rr], d=r64[1].numpy()[rr], o=r64[0].numpy()[rr], inv_s=inv_s, cos_anneal=cos_anneal,
                     inside_hip=res["inside"].cpu().numpy()[rr], z=z.numpy()[rr])
        sdf = out["sdf"].numpy()
        inv_s = float(orc.inv_s_of(params))
        for r in worst:
            j = int(np.argmax(dw[r]))
            jc = int(np.argmax(dc[r] * w[r]))
            print(f"   ray {r}: rgb diff {drgb[r]:.3e}; sum w hip {w_hip[r].sum():.7f} oracle {w[r].sum():.7f}; worst weight diff {dw[r, j]:.3e} at "
                  f"sample {j} (w {w[r, j]:.5f}, sdf*s {sdf[r, j] * inv_s:+.3f}, dist*s {float(dist[r, j]) * inv_s:.4f}, alpha {float(out['alpha'][r, j]):.6f}); "
                  f"worst weighted colour diff {dc[r, jc] * w[r, jc]:.3e} at sample {jc} (colour diff {dc[r, jc]:.3e}, w {w[r, jc]:.4f}); cue {cue[r].numpy()}")


if __name__ == "__main__":
    main()
