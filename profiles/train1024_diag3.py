#!/usr/bin/env python3
"""Which part of the f16x3 path carries the remaining 1.3e-4-of-scale error of the SDF value-path gradients at global_step 25 000
(profiles/r05/train1024_diag_scaled.log)?  Variants of the fused step: wide (one-wave-per-SIMD) no-grad kernels on / off, shadow
march's last evaluation in forward mode (no unorm16 sigma')."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nrhints_amd as na
from nrhints_amd import train_fused
from nrhints_amd.synthetic import perturb_state
from tests.conftest import load_npz

T = torch.from_numpy
cu = lambda a: T(np.asarray(a)).float().contiguous().cuda()
g = load_npz("train1024_b.npz")
state = perturb_state(load_npz("scene_a_state.npz"))
names = ("sdf_network.lin7.weight_g", "sdf_network.out_sdf.weight_g", "sdf_network.out_sdf.bias", "sdf_network.lin5.weight_g",
         "sdf_network.out_sdf.weight_v", "deviation_network.variance", "color_network.lin0.weight_g")
for gs in (25000, 100000):
    p = f"s{gs}."
    for tag, prec, kw in (("f32", "f32", {}), ("f16x3", "f16x3", {}), ("f16x3 16-point samplers", "f16x3", dict(wide_kernels=False)),
                          ("f16x3 shadow jvp", "f16x3", dict(shadow_jvp=True))):
        m = na.NeuSHintRenderer(na.NeuSModelConfig(), precision=prec)
        for k, v in kw.items():
            setattr(m, k, v)
        m.load_state_dict({k: T(np.asarray(v)) for k, v in state.items()})
        m = m.cuda()
        rb = na.RayBundle(origins=cu(g["o"]), directions=cu(g["d"]), pl_positions=cu(g["pl"]), nears=cu(g["near"]), fars=cu(g["far"]))
        train_fused.train_step_backward(m, rb, cu(g["rgb_gt"]), torch.ones(1, 3).cuda(), gs, t_rand_primary=cu(g[p + "t_rand_primary"]),
                                        t_rand_shadow=cu(g[p + "t_rand_shadow"]))
        prm = dict(m.named_parameters())
        B = next(iter(m._fused_buffers.values()))
        rgb_err = float(np.abs(B.rgb.cpu().numpy() - g[p + "rgb_f64"]).max())
        out = []
        for n_ in names:
            want = g[p + "grad64." + n_].astype(np.float64)
            err = np.abs(prm[n_].grad.detach().cpu().numpy().astype(np.float64) - want).max() / max(np.abs(want).max(), 1e-12)
            out.append(f"{err:.1e}")
        print(f"step {gs} {tag:26s} rgb err {rgb_err:.1e} | " + " ".join(out))
print("columns:", " ".join(n_.replace("_network", "") for n_ in names))
