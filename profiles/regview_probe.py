#!/usr/bin/env python3
"""Where does a register_view step spend its time? (measurement aid)  cProfile of 60 fused steps + wall clock per step."""
import cProfile, io, os, pstats, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, nrhints_amd as na
from nrhints_amd import RawPixelBundle, RayGenerator, RayGeneratorConfig
from nrhints_amd.pipeline import CameraModel
from nrhints_amd.training import register_view

dev = torch.device("cuda", 0)
ncam, Hc, Wc = 12, 200, 200
cam = CameraModel(H=Hc, W=Wc, cx=Wc / 2, cy=Hc / 2, fx=280.0, fy=280.0)
poses, pls = bench.orbit_views(ncam)
model, _ = bench.build_scene("f16x3")
model = model.to(dev).eval()
hh, ww = np.meshgrid(np.arange(Hc, dtype=np.float32), np.arange(Wc, dtype=np.float32), indexing="ij")
view = 3
img = RawPixelBundle(img_indices=torch.full((Hc, Wc, 1), view, dtype=torch.long), h_indices=torch.from_numpy(hh)[..., None],
                     w_indices=torch.from_numpy(ww)[..., None], poses=torch.from_numpy(poses[view]).expand(Hc, Wc, 4, 4),
                     pls=torch.from_numpy(pls[view]).expand(Hc, Wc, 3), rgb_gt=torch.rand(Hc, Wc, 3))
rg = RayGenerator(cam, ncam, RayGeneratorConfig(cam_opt_mode="SO3xR3")).to(dev)
gen = torch.Generator().manual_seed(1)
register_view(model, rg, img, dev, steps=10, batch_size=512, lr=1e-3, generator=gen)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
register_view(model, rg, img, dev, steps=60, batch_size=512, lr=1e-3, generator=gen)
torch.cuda.synchronize()
pr.disable()
print("ms per step:", (time.perf_counter() - t0) / 60 * 1e3)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
