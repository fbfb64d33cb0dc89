#!/usr/bin/env python3
"""VERDICT r4 item 6 (i), priced on the CPU before any kernel is written: would a TWO-term product (one cross term of the f16x3
split dropped) in the sampler passes (sdf32_kernel<0>: 25 % of a frame) stay inside the evaluation gate (rgb within 3e-5 of the
reference's recorded render, tests/test_gpu_parity.py::_check_against)?

Dropping A_lo * B_hi  = the WEIGHTS rounded to fp16 (11 bits) in those passes (also halves the weight stream);
dropping A_hi * B_lo  = the ACTIVATIONS rounded to fp16 after every softplus.
Either way only the SAMPLER's SDF evaluations change (both samplers: primary and shadow ray); the sections they produce are then
rendered with the exact network, as the kernels would.  Float64 oracle, quantisation applied where the kernel would apply it
(scaled softplus domain for activations: u = 100 h / ln 2), on the reference's own recorded rays (tests/golden/render_*.npz).

    python profiles/two_term_sampler_emulation.py  >  profiles/r05/two_term_sampler_emulation.log
"""
import math, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import neus_oracle as orc
from nrhints_amd.synthetic import perturb_state, psnr

T = torch.from_numpy
IK = 100.0 / math.log(2.0)
f16 = lambda t: t.to(torch.float16).to(t.dtype)


def quant_sdf_forward(mode):
    real = orc.sdf_forward

    def fwd(p, pts, want_feat=True):
        e = orc.nerf_encode(pts * 3.0, 6)
        h = e
        for l in range(8):
            if l == 4:
                h = torch.cat([h, e], dim=1) / math.sqrt(2.0)
            w = f16(p.sdf_w[l]) if mode == "weights" else p.sdf_w[l]
            x = h
            if mode == "acts":
                x = f16(h * IK) / IK if l > 0 else f16(h)          # the kernel's B operand: u = 100 h / ln 2 (L0: the raw embedding)
            h = orc.softplus100(torch.nn.functional.linear(x, w, p.sdf_b[l]))
        hw = f16(p.sdf_head_w) if mode == "weights" else p.sdf_head_w
        x = f16(h * IK) / IK if mode == "acts" else h
        return torch.nn.functional.linear(x, hw, p.sdf_head_b) / 3.0, None
    return real, fwd


def main():
    a = dict(np.load(os.path.join(ROOT, "tests", "golden", "scene_a_state.npz")))
    for tag, state in (("a", a), ("b", perturb_state(a))):
        g = dict(np.load(os.path.join(ROOT, "tests", "golden", f"render_{tag}.npz")))
        rays = [T(g[k]).double() for k in ("o", "d", "pl", "near", "far")]
        p = orc.params_from_state(state, dtype=torch.float64)
        base = orc.render_forward(p, *rays, background_rgb=torch.ones(1, 3, dtype=torch.float64), mode="minimal")
        ref = g["rgb_f64"]
        print(f"scene {tag}: exact sampler, float64 oracle vs the reference's float64 record: max |rgb| {np.abs(base['rgb'].numpy() - ref).max():.2e}")
        real_h = orc.hierarchical_z
        for mode in ("weights", "acts"):
            real, fwd = quant_sdf_forward(mode)

            def hz(pp, o, d, z, n_steps=4, n_new=16, full_forward=True):
                orc.sdf_forward = fwd
                try:
                    return real_h(pp, o, d, z, n_steps, n_new, False)
                finally:
                    orc.sdf_forward = real
            orc.hierarchical_z = hz
            try:
                out = orc.render_forward(p, *rays, background_rgb=torch.ones(1, 3, dtype=torch.float64), mode="minimal")
            finally:
                orc.hierarchical_z = real_h
            # sdf error of the quantised evaluation itself, on the coarse samples
            pts = (rays[0][:, None] + rays[1][:, None] * torch.linspace(0, 1, 64, dtype=torch.float64)[None, :, None] * 2 + rays[1][:, None] * (rays[3][:, None] - 0)).reshape(-1, 3)
            es = (fwd(p, pts)[0] - real(p, pts, False)[0]).abs().max().item()
            d_rgb = np.abs(out["rgb"].numpy() - ref)
            print(f"scene {tag}: two-term sampler, {mode:7s} at fp16: sampler sdf error max {es:.2e}; rgb vs reference max {d_rgb.max():.2e} "
                  f"mean {d_rgb.mean():.2e}, PSNR {psnr(out['rgb'].numpy(), ref):.1f} dB; visibility max {np.abs(out['visibilities'].numpy() - g['visibilities_f64']).max():.2e}"
                  f"  -> gate rgb <= 3e-5: {'PASS' if d_rgb.max() <= 3e-5 else 'FAIL'}")


if __name__ == "__main__":
    main()
