#!/usr/bin/env python3
"""Write the flat binary scene file examples/c_abi_render.cpp reads: the packed network buffers of a NeuSHintRenderer
(host-side packing, nrhints_amd/packing.py) and a batch of rays.  CPU only - no GPU needed to produce the file.

    python examples/dump_scene.py scene.bin [nrays] [precision]
"""
import os
import struct
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nrhints_amd as na  # noqa: E402
from nrhints_amd import _lib, packing, packing32  # noqa: E402
from nrhints_amd.synthetic import make_rays  # noqa: E402

MAGIC = 0x4e52483031


def dump(path, model, rays, background=(1.0, 1.0, 1.0), cos_anneal=1.0):
    """model: NeuSHintRenderer; rays: (o, d, pl, near, far) numpy float32.  The weight-norm fold and the packing run on the
    model's device with the fold the renderer itself uses there (packing.dense_params_device: a GPU-resident model gives
    bit-identical buffers to the ones its own forward() uses; the CPU fold differs from the GPU kernel's in last bits)."""
    state = {k: v.detach().float() for k, v in model.state_dict().items()}
    d = packing.dense_params_device(state)
    prec = _lib.PRECISIONS[model.precision]
    hints = bool(model._hints)
    sw, sb, sh = packing.pack_sdf(d, prec)
    cw, cb = packing.pack_color(d, prec, hints)
    inv_s = float(torch.exp(state["deviation_network.variance"] * 10.0).clip(1e-6, 1e6))
    blobs = [t.contiguous().cpu().numpy().tobytes() for t in (sw, sb, sh, cw, cb)]
    wide = b""
    if prec == 1 and getattr(model, "wide_kernels", True):
        # f16x3: the streams and tables of the wide SDF kernels (NrhNet.sdf_w32 / sdf_tab32) behind the five classic buffers
        # with the feature head multiplied into the reflectance net's first layer (NrhNet.feat_fused), as the renderer does
        w32, tab32 = packing32.pack_sdf32_fused(d)
        wide = w32.contiguous().cpu().numpy().tobytes() + tab32.contiguous().cpu().numpy().tobytes()
        if hints and getattr(model, "wide_color", True):     # + the reflectance net for the wide kernel (NrhNet.col_w32 / col_tab32)
            c32, ctab = packing32.pack_color32(d)
            wide += c32.contiguous().cpu().numpy().tobytes() + ctab.contiguous().cpu().numpy().tobytes()
    o, dr, pl, near, far = (np.ascontiguousarray(a, dtype=np.float32) for a in rays)
    n = o.shape[0]
    with open(path, "wb") as f:
        f.write(struct.pack("<10q", MAGIC, prec, int(hints) | (2 if wide else 0), n, *[len(b) for b in blobs], len(wide)))
        f.write(struct.pack("<2f", inv_s, cos_anneal))
        for b in blobs:
            f.write(b)
        f.write(wide)
        for a in (o, dr, pl, near.reshape(-1), far.reshape(-1), np.asarray(background, dtype=np.float32),
                  torch.linspace(0.0, 1.0, 64).numpy(), torch.linspace(0.0, 1.0, 16).numpy()):
            f.write(np.ascontiguousarray(a, dtype=np.float32).tobytes())
    return n


if __name__ == "__main__":
    out = sys.argv[1]
    nrays = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    precision = sys.argv[3] if len(sys.argv) > 3 else "f16x3"
    torch.manual_seed(0)
    m = na.NeuSHintRenderer(na.NeuSModelConfig(), precision=precision)
    print("wrote", dump(out, m, make_rays(nrays, seed=0, spread=0.1)), "rays to", out)
