#!/usr/bin/env python3
"""Localise differences between csrc/nrh_color32.hip and its numpy emulation (tests/mfma32_emulator.color32_tile): the same
packed stream, arbitrary inputs, one input group switched on at a time."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np, torch
from nrhints_amd import ops, packing as pk, packing32 as pk32
from tests import mfma32_emulator as emu

st = dict(np.load(os.path.join(ROOT, "tests/golden/scene_a_state.npz")))
d = pk.dense_params({k: torch.from_numpy(np.asarray(v)).cuda() for k, v in st.items()})
c32, ctab = pk32.pack_color32(d)
c32n, ctabn = c32.cpu().numpy(), ctab.cpu().numpy()
N = 2
g = torch.Generator().manual_seed(3)
base = dict(o=torch.randn(N, 3, generator=g) * 0.3, d=torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1),
            t=torch.rand(N, 128, generator=g), n=torch.nn.functional.normalize(torch.randn(N * 128, 3, generator=g), dim=-1),
            part=torch.randn(N * 128, 256, generator=g) * 0.3, rm=torch.randn(N + 1, 100, generator=g))
base["rm"][:, 99] = 0

def run(cfg):
    x = {k: v.clone() for k, v in base.items()}
    if "nopart" in cfg: x["part"].zero_()
    if "norm" in cfg: x["rm"].zero_()
    if "nopn" in cfg: x["o"].zero_(); x["d"].zero_(); x["n"].zero_()
    for s in range(1, 8):
        if f"only{s}" in cfg:
            keep = x["rm"][:, 16 * (s - 1):16 * s].clone() if s < 7 else x["rm"][:, 96:99].clone()
            x["rm"].zero_()
            if s < 7: x["rm"][:, 16 * (s - 1):16 * s] = keep
            else: x["rm"][:, 96:99] = keep
    col = ops.color_eval_wide(c32, ctab, pk.rows_to_feat_tiles(x["part"]).cuda(), x["o"].cuda(), x["d"].cuda(), x["t"].cuda(),
                              x["n"].cuda().contiguous(), x["rm"].cuda()).cpu().numpy().reshape(N, 128, 3)
    worst = 0.0
    per_tile = []
    for ray in range(N):
        for w in range(4):
            sl = slice(32 * w, 32 * w + 32)
            pts = (x["o"][ray] + x["d"][ray] * x["t"][ray, sl, None]).double().numpy()
            ref = emu.color32_tile(c32n, ctabn, x["part"][ray * 128 + 32 * w: ray * 128 + 32 * w + 32].double().numpy(), pts,
                                   x["n"][ray * 128 + 32 * w: ray * 128 + 32 * w + 32].double().numpy(), x["rm"][ray].double().numpy())
            e = float(np.abs(col[ray, sl] - ref).max())
            per_tile.append(round(e, 7))
            worst = max(worst, e)
    print(f"{cfg:28s} max err {worst:.3e}   per tile {per_tile}", flush=True)

for cfg in ("all", "nopart", "norm", "nopn", "nopart norm", "nopart norm nopn", "nopart nopn only1", "nopart nopn only4", "nopart nopn only7"):
    run(cfg)
