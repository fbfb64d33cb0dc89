#!/usr/bin/env python3
"""VERDICT r4 "missing" 5, priced on the CPU: what would a SINGLE-pass fp16 MFMA mode (one product term: weights AND activations of the SDF
network at fp16's 11 bits, fp32 accumulation; ceiling 3.3 M rays/s) do to the picture?  The gate SURVEY 8c / 8d gives a
reduced-precision mode is PSNR(ours, reference) >= 50 dB (keeps |delta PSNR vs ground truth| < 0.05 dB at 30 dB).

Float64 oracle in the reference's call pattern (mode "as_written": autograd gradient, so d sdf / dx is the derivative of the QUANTISED
network, as a one-term reverse chain would compute it), every SDF-network evaluation of the render quantised - both samplers, render_core,
the shadow march - with the operand roundings where the wide kernels would apply them (scaled softplus domain u = 100 h / ln 2 for
activations); the reflectance net stays exact (it is the benign one) - and, last block (round 5, after the SDF kernels were
built: what the NEXT step of the mode would cost), ALSO at one term: weights and layer inputs of the reflectance net at fp16.  Scenes a (1/s ~ 20) and b (1/s ~ 1 100, trained-like) on
the reference's recorded rays (tests/golden/render_*.npz).

    python profiles/one_term_emulation.py  >  profiles/r05/one_term_emulation.log
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
bf16 = lambda t: t.to(torch.bfloat16).to(t.dtype)


def make_forward(q):
    def fwd(p, pts, want_feat=True):
        e = orc.nerf_encode(pts * 3.0, 6)
        h = e
        for l in range(8):
            if l == 4:
                h = torch.cat([h, e], dim=1) / math.sqrt(2.0)
            x = q(h * IK) / IK if l > 0 else q(h)
            h = orc.softplus100(torch.nn.functional.linear(x, q(p.sdf_w[l]), p.sdf_b[l]))
        x = q(h * IK) / IK
        sdf = torch.nn.functional.linear(x, q(p.sdf_head_w), p.sdf_head_b) / 3.0
        feat = torch.nn.functional.linear(x, q(p.feat_w), p.feat_b) if want_feat else None
        return sdf, feat
    return fwd


def make_color(q):
    def col(p, pts, normals, view, feat, pls, vis=None, cue=None):
        parts = [pts, orc.nerf_encode(view, 4), normals, orc.nerf_encode(pls, 4), feat]
        if vis is not None:
            parts.append(orc.nerf_encode(vis, 4))
        if cue is not None:
            parts.append(orc.nerf_encode(cue, 4))
        x = torch.cat(parts, dim=-1)
        for l in range(5):
            x = torch.nn.functional.linear(q(x), q(p.col_w[l]), p.col_b[l])
            if l < 4:
                x = torch.relu(x)
        return torch.sigmoid(x)
    return col


def main():
    a = dict(np.load(os.path.join(ROOT, "tests", "golden", "scene_a_state.npz")))
    real = orc.sdf_forward
    for tag, state in (("a", a), ("b", perturb_state(a))):
        g = dict(np.load(os.path.join(ROOT, "tests", "golden", f"render_{tag}.npz")))
        rays = [T(g[k]).double() for k in ("o", "d", "pl", "near", "far")]
        p = orc.params_from_state(state, dtype=torch.float64)
        ref = g["rgb_f64"]
        for name, q in (("fp16 (one-term f16 MFMA)", f16), ("bf16 (one-term bf16 MFMA)", bf16)):
            orc.sdf_forward = make_forward(q)
            try:
                out = orc.render_forward(p, *rays, background_rgb=torch.ones(1, 3, dtype=torch.float64), mode="as_written")
            finally:
                orc.sdf_forward = real
            d = np.abs(out["rgb"].numpy() - ref)
            ps = psnr(out["rgb"].numpy(), ref)
            print(f"scene {tag} (1/s = {float(orc.inv_s_of(p)):.0f}), {name}: rgb max {d.max():.2e} mean {d.mean():.2e}  PSNR(ours, reference) {ps:.1f} dB  "
                  f"depth max {np.abs(out['depth'].numpy() - g['depth_f64']).max():.2e}  visibility max {np.abs(out['visibilities'].numpy() - g['visibilities_f64']).max():.2e}"
                  f"  -> gate >= 50 dB: {'PASS' if ps >= 50.0 else 'FAIL'}")
        # ... and with the reflectance net at one term as well
        real_c = orc.color_forward
        orc.sdf_forward, orc.color_forward = make_forward(f16), make_color(f16)
        try:
            out = orc.render_forward(p, *rays, background_rgb=torch.ones(1, 3, dtype=torch.float64), mode="as_written")
        finally:
            orc.sdf_forward, orc.color_forward = real, real_c
        d = np.abs(out["rgb"].numpy() - ref)
        ps = psnr(out["rgb"].numpy(), ref)
        print(f"scene {tag}, fp16 one-term SDF net AND reflectance net: rgb max {d.max():.2e} mean {d.mean():.2e}  PSNR(ours, reference) {ps:.1f} dB"
              f"  -> gate >= 50 dB: {'PASS' if ps >= 50.0 else 'FAIL'}", flush=True)


if __name__ == "__main__":
    main()
