#!/usr/bin/env python3
"""Kernel micro-benchmark: time the MLP kernels of one libnrhints_hip.so build (NRHINTS_HIP_LIB) in isolation.

    NRHINTS_HIP_LIB=/path/to/variant.so python profiles/kbench.py [nrays] [precision ...]
Prints one line per (precision, kernel): ms per launch and algorithmic TFLOP/s.  Used to A/B compile-time knobs
(csrc/nrh_mlp.h NRH_*) within one gpurun call; every variant sees the same seeded inputs."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nrhints_amd as na
from nrhints_amd import ops, packing as pk, _lib
from nrhints_amd.synthetic import make_rays, perturb_state

MAC_F_FULL, MAC_F_SDF, MAC_G, MAC_C = 524_544, 459_008, 459_008, 289_792
FLOP = {0: 2 * MAC_F_SDF, 1: 2 * (MAC_F_SDF + MAC_G), 2: 2 * (MAC_F_FULL + MAC_G), "color": 2 * MAC_C}

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

def main():
    nrays = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    precs = sys.argv[2:] or ["f16x3", "f32"]
    tag = os.path.basename(os.environ.get("NRHINTS_HIP_LIB", "default"))
    torch.manual_seed(0)
    base = na.NeuSHintRenderer()
    st = perturb_state({k: v.detach().numpy().copy() for k, v in base.state_dict().items()})
    o, d, pl, near, far = (torch.from_numpy(a).cuda() for a in make_rays(nrays, seed=1, spread=0.1))
    t = (near + (far - near) * torch.linspace(0, 1, 128, device="cuda")[None]).contiguous()
    npts = nrays * 128
    for prec in precs:
        m = na.NeuSHintRenderer(precision=prec)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
        p = m.cuda().packed_params(torch.device("cuda", 0))
        scratch = ops._scratch(o.device)
        res = {}
        for mode in (0, 1, 2):
            f = lambda: ops.sdf_eval(mode, p["sdf_w"], p["sdf_b"], p["sdf_head"], o, d, t, 128, scratch=scratch)
            ms = timeit(f)
            res[f"sdf{mode}"] = (ms, FLOP[mode] * npts / ms / 1e9)
        sdf, grad, feat = ops.sdf_eval(2, p["sdf_w"], p["sdf_b"], p["sdf_head"], o, d, t, 128, scratch=scratch)
        nhat = torch.nn.functional.normalize(grad, dim=-1).contiguous()
        raymisc = torch.rand(nrays, pk.RAYMISC_STRIDE, device="cuda")
        f = lambda: ops.color_eval(p["col_w"], p["col_b"], feat, o, d, t, nhat, raymisc)
        ms = timeit(f)
        res["color"] = (ms, FLOP["color"] * npts / ms / 1e9)
        chk = float(sdf.double().sum().item()), float(grad.double().abs().sum().item())
        print(f"{tag:28s} {prec:6s} " + " ".join(f"{k}={v[0]:7.2f}ms/{v[1]:6.1f}TF" for k, v in res.items()) +
              f"  chk={chk[0]:.6f},{chk[1]:.4f}", flush=True)

if __name__ == "__main__":
    main()
