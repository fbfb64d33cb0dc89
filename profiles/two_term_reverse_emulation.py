#!/usr/bin/env python3
"""A follow-up to profiles/two_term_sampler_emulation.py, priced on the CPU (nothing built): a TWO-term product in the REVERSE
chain only - the eight W_l^T stages of sdf32_kernel<1> / <2> that turn sigma' into d sdf / dx (the analytic normal).  The sampler's
decisions and the SDF values stay three-term, so no sample moves; only the normal changes - what the reflectance net reads, the
cos in alpha, the cue, the shadow ray's alpha.  Those stages are 8 of the 17 stage-equivalents of sdf32_kernel<2> (the frame's
dominant kernel, 35 % of it) and 8 of 16 of sdf32_kernel<1> (the shadow rays): -1/6 of their MFMAs.

  weights   W_l^T rounded to fp16 in the reverse chain (drops A_lo * B_hi; also halves those stages' weight stream)
  acts      the chain's running vector (g * sigma') rounded to fp16 (drops A_hi * B_lo)

Gate: rgb within 3e-5 of the reference's recorded float64 render (tests/test_gpu_parity.py::_check_against), as for the sampler.

    python profiles/two_term_reverse_emulation.py  >  profiles/r05/two_term_reverse_emulation.log
"""
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import neus_oracle as orc  # noqa: E402
from nrhints_amd.synthetic import perturb_state, psnr  # noqa: E402

T = torch.from_numpy
f16 = lambda t: t.to(torch.float16).to(t.dtype)  # noqa: E731


def make(mode):
    def fga(p, pts, want_feat=True):
        x3 = pts * 3.0
        e = orc.nerf_encode(x3, 6)
        h = e
        dact = []
        for l in range(8):
            if l == 4:
                h = torch.cat([h, e], dim=1) / math.sqrt(2.0)
            z = F.linear(h, p.sdf_w[l], p.sdf_b[l])
            t = z * 100.0
            ez = torch.exp(t)
            dact.append(torch.where(t > 20.0, torch.ones_like(t), ez / (ez + 1.0)))
            h = orc.softplus100(z)
        sdf = F.linear(h, p.sdf_head_w, p.sdf_head_b) / 3.0
        feat = F.linear(h, p.feat_w, p.feat_b) if want_feat else None
        g = (p.sdf_head_w / 3.0).expand(pts.shape[0], -1)
        ge_skip = None
        for l in range(7, -1, -1):
            v = g * dact[l]
            w = p.sdf_w[l]
            if mode == "weights":
                w = f16(w)
            elif mode == "acts":
                # the kernel's operand carries a per-layer power-of-two scale that keeps it in fp16's normal range: emulate with one
                s = 2.0 ** (10 - math.ceil(math.log2(float(v.abs().max()) + 1e-300)))
                v = f16(v * s) / s
            g = v @ w
            if l == 4:
                g = g / math.sqrt(2.0)
                ge_skip = g[:, 217:]
                g = g[:, :217]
        ge = g + ge_skip
        freqs = 2.0 ** torch.linspace(0.0, 5.0, 6, dtype=pts.dtype)
        s = (x3[..., None] * freqs)
        gs = ge[:, 3:21].reshape(-1, 3, 6)
        gc = ge[:, 21:39].reshape(-1, 3, 6)
        dx3 = ge[:, 0:3] + ((gs * torch.cos(s) + gc * torch.cos(s + math.pi / 2.0)) * freqs).sum(-1)
        return sdf, feat, dx3 * 3.0
    return fga


def main():
    a = dict(np.load(os.path.join(ROOT, "tests", "golden", "scene_a_state.npz")))
    for tag, state in (("a", a), ("b", perturb_state(a))):
        g = dict(np.load(os.path.join(ROOT, "tests", "golden", f"render_{tag}.npz")))
        rays = [T(g[k]).double() for k in ("o", "d", "pl", "near", "far")]
        p = orc.params_from_state(state, dtype=torch.float64)
        ref = g["rgb_f64"]
        real = orc.sdf_forward_grad_analytic
        base = orc.render_forward(p, *rays, background_rgb=torch.ones(1, 3, dtype=torch.float64), mode="minimal")
        print(f"scene {tag}: exact reverse chain, float64 oracle vs the reference's float64 record: max |rgb| {np.abs(base['rgb'].numpy() - ref).max():.2e}")
        for mode in ("weights", "acts"):
            orc.sdf_forward_grad_analytic = make(mode)
            try:
                out = orc.render_forward(p, *rays, background_rgb=torch.ones(1, 3, dtype=torch.float64), mode="minimal")
            finally:
                orc.sdf_forward_grad_analytic = real
            d_rgb = np.abs(out["rgb"].numpy() - ref)
            dn = (out["normal"].numpy() - base["normal"].numpy()) if "normal" in out else None
            nerr = f"{np.abs(dn).max():.2e}" if dn is not None else "n/a"
            print(f"scene {tag}: two-term reverse chain, {mode:7s} at fp16: rgb vs reference max {d_rgb.max():.2e} mean {d_rgb.mean():.2e}, "
                  f"PSNR {psnr(out['rgb'].numpy(), ref):.1f} dB; depth max {np.abs(out['depth'].numpy() - base['depth'].numpy()).max():.2e}; "
                  f"visibility max {np.abs(out['visibilities'].numpy() - g['visibilities_f64']).max():.2e}; normal map max {nerr}"
                  f"  -> gate rgb <= 3e-5: {'PASS' if d_rgb.max() <= 3e-5 else 'FAIL'}", flush=True)


if __name__ == "__main__":
    main()
