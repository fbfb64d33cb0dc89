#!/usr/bin/env python3
"""Where does the one-hint model's first-reflectance-layer gradient differ from the reference's float64 run?  (debug aid)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nrhints_amd as na
from nrhints_amd.synthetic import one_hint_state, perturb_state
from nrhints_amd.training import train_loss_dict
T = torch.from_numpy
g = dict(np.load(os.path.join(ROOT, "tests/golden/render_branches_b.npz")))
sb = perturb_state(dict(np.load(os.path.join(ROOT, "tests/golden/scene_a_state.npz"))))
cu = lambda a: T(np.asarray(a)).float().contiguous().cuda()
for vt, shadow in (("sho", True), ("spo", False)):
    st = one_hint_state(sb, shadow)
    R = na.NeuSRendererConfig
    res = {}
    for prec in ("f32", "f16x3"):
        for rep in range(2):
            m = na.NeuSHintRenderer(na.NeuSModelConfig(renderer=R(shadow_hint=shadow, specular_hint=not shadow)), precision=prec)
            m.load_state_dict({k: T(np.asarray(v)) for k, v in st.items()})
            m = m.cuda()
            a = [cu(g["t." + k]) for k in ("o", "d", "pl", "near", "far")]
            rb = na.RayBundle(origins=a[0], directions=a[1], pl_positions=a[2], nears=a[3], fars=a[4])
            out = m(rb, is_training=True, background_rgb=torch.ones(1, 3).cuda(), global_step=int(g["t.global_step"]),
                    _t_rand_primary=cu(g[f"{vt}.t_rand_primary"]), _t_rand_shadow=cu(g[f"{vt}.t_rand_shadow"]) if shadow else None)
            train_loss_dict(out, cu(g["t.rgb_gt"]), 0.1)["loss"].backward()
            res[(prec, rep)] = {k: p.grad.detach().cpu().numpy().astype(np.float64) for k, p in m.named_parameters()}
    for name in ("color_network.lin0.weight_v", "color_network.lin0.weight_g"):
        w64, w32 = g[f"{vt}.grad64.{name}"], g[f"{vt}.grad.{name}"]
        sc = np.abs(w64).max()
        print(vt, name, "scale", sc, "ref32-64", np.abs(w32 - w64).max() / sc)
        for key, r in res.items():
            d = np.abs(r[name] - w64) / sc
            line = f"   {key}: max {d.max():.2e}"
            if d.ndim == 2 and d.shape[1] > 300:
                blocks = {"pts": (0, 3), "view": (3, 30), "nrm": (30, 33), "pl": (33, 60), "feat": (60, 316), "hint": (316, d.shape[1])}
                line += "  " + " ".join(f"{b}:{d[:, lo:hi].max():.1e}" for b, (lo, hi) in blocks.items())
                i = np.unravel_index(d.argmax(), d.shape)
                line += f"  argmax {i} got {r[name][i]:.3e} want {w64[i]:.3e}"
            print(line)
        print("   run-to-run f16x3:", np.abs(res[("f16x3", 0)][name] - res[("f16x3", 1)][name]).max() / sc)

# the full two-hint model on the same rays: f16x3 against f32 (no reference needed) - is the pl block's sensitivity a property of the
# one-hint padding or of the precision mode?
res = {}
for prec in ("f32", "f16x3"):
    m = na.NeuSHintRenderer(na.NeuSModelConfig(), precision=prec)
    m.load_state_dict({k: T(np.asarray(v)) for k, v in sb.items()})
    m = m.cuda()
    a = [cu(g["t." + k]) for k in ("o", "d", "pl", "near", "far")]
    rb = na.RayBundle(origins=a[0], directions=a[1], pl_positions=a[2], nears=a[3], fars=a[4])
    out = m(rb, is_training=True, background_rgb=torch.ones(1, 3).cuda(), global_step=int(g["t.global_step"]),
            _t_rand_primary=cu(g["shg.t_rand_primary"]), _t_rand_shadow=cu(g["shg.t_rand_shadow"]))
    train_loss_dict(out, cu(g["t.rgb_gt"]), 0.1)["loss"].backward()
    res[prec] = ({k: p.grad.detach().cpu().numpy().astype(np.float64) for k, p in m.named_parameters()},
                 out.rgb.detach().cpu().numpy(), out.weights.detach().cpu().numpy(), out.visibilities.detach().cpu().numpy())
name = "color_network.lin0.weight_v"
a, b = res["f32"][0][name], res["f16x3"][0][name]
sc = np.abs(a).max()
d = np.abs(a - b) / sc
blocks = {"pts": (0, 3), "view": (3, 30), "nrm": (30, 33), "pl": (33, 60), "feat": (60, 316), "vis": (316, 325), "cue": (325, 361)}
print("full model f16x3 vs f32:", name, "max", d.max(), " ".join(f"{k}:{d[:, lo:hi].max():.1e}" for k, (lo, hi) in blocks.items()))
print("   rgb diff", np.abs(res["f32"][1] - res["f16x3"][1]).max(), "weights diff max/mean", np.abs(res["f32"][2] - res["f16x3"][2]).max(),
      np.abs(res["f32"][2] - res["f16x3"][2]).mean(), "vis diff", np.abs(res["f32"][3] - res["f16x3"][3]).max())
for k in ("sdf_network.lin0.weight_v", "sdf_network.lin7.bias", "color_network.lin2.bias", "deviation_network.variance"):
    a, b = res["f32"][0][k], res["f16x3"][0][k]
    print("   ", k, np.abs(a - b).max() / max(np.abs(a).max(), 1e-30))
