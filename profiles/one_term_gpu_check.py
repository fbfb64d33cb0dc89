#!/usr/bin/env python3
"""precision "f16" (the one-term builds of the wide SDF kernels) against the reference's recorded renders (tests/golden/render_*.npz,
96 rays each) and against the float64 oracle on a strided sample of the benchmark frame: PSNR, max error - the numbers CHANGELOG.md section 7h quotes."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nrhints_amd as na
from nrhints_amd.synthetic import make_image_rays, perturb_state, psnr
from tests.conftest import load_npz

T = torch.from_numpy
a = load_npz("scene_a_state.npz")
for tag, st in (("a", a), ("b", perturb_state(a))):
    g = load_npz(f"render_{tag}.npz")
    rb = na.RayBundle(origins=T(g["o"]).cuda(), directions=T(g["d"]).cuda(), pl_positions=T(g["pl"]).cuda(), nears=T(g["near"]).cuda(), fars=T(g["far"]).cuda())
    row = []
    for prec in ("f16", "f16x3", "f32"):
        m = na.NeuSHintRenderer(na.NeuSModelConfig(), precision=prec)
        m.load_state_dict({k: T(np.asarray(v)) for k, v in st.items()})
        m = m.cuda().eval()
        with torch.no_grad():
            out = m(rb, background_rgb=torch.ones(1, 3).cuda())
        rgb = out.rgb.cpu().numpy()
        row.append(f"{prec}: PSNR {psnr(rgb, g['rgb_f64']):.1f} dB, max |rgb| {np.abs(rgb - g['rgb_f64']).max():.1e}, depth {np.abs(out.depth.cpu().numpy() - g['depth_f64']).max():.1e}, "
                   f"visibility {np.abs(out.visibilities.cpu().numpy() - g['visibilities_f64']).max():.1e}")
    print(f"scene {tag} vs the reference's float64 render (96 rays): " + " | ".join(row))
# the benchmark frame: f16 against f16x3 on every 16th row
st = perturb_state(a)
rays = make_image_rays(800, 800, azimuth=0.6, elevation=0.5)
sel = np.arange(0, 800 * 800, 1).reshape(800, 800)[::16].reshape(-1)
rb = na.RayBundle(**{k: T(v[sel]).cuda() for k, v in zip(("origins", "directions", "pl_positions", "nears", "fars"), rays)})
imgs = {}
for prec in ("f16", "f16x3"):
    m = na.NeuSHintRenderer(na.NeuSModelConfig(), precision=prec)
    m.load_state_dict({k: T(np.asarray(v)) for k, v in st.items()})
    m = m.cuda().eval()
    with torch.no_grad():
        imgs[prec] = m(rb, background_rgb=torch.ones(1, 3).cuda()).rgb.cpu().numpy()
d = np.abs(imgs["f16"] - imgs["f16x3"])
print(f"benchmark frame, every 16th row ({len(sel)} rays): f16 vs f16x3 PSNR {psnr(imgs['f16'], imgs['f16x3']):.1f} dB, max {d.max():.2e}, "
      f"99.9th percentile {np.percentile(d, 99.9):.2e}, mean {d.mean():.2e}, pixels off by > 1e-2: {(d.max(1) > 1e-2).sum()}")
