#!/usr/bin/env python3
"""Test vectors for profiles/ubench/sdf32_bench.hip: packed streams + tables of synthetic scene b, rays, sample positions and
the fp64 oracle's sdf / gradient / feature at the first NCHECK points.  Output: profiles/ubench/bin/sdf32_case.bin
(little-endian: header of int64 counts, then the arrays).  Runs on CPU; the file travels to the GPU box with the snapshot."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nrhints_amd as na
from nrhints_amd import packing as pk, packing32 as pk32
from nrhints_amd.synthetic import make_rays, perturb_state
from oracle import neus_oracle as orc

def main():
    nrays = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    nper = 128
    ncheck = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    scene = sys.argv[2] if len(sys.argv) > 2 else 'b'
    torch.manual_seed(0)
    base = na.NeuSHintRenderer()
    st = {k: v.detach().numpy().copy() for k, v in base.state_dict().items()}
    if scene == 'b':
        st = perturb_state(st)
    d = pk.dense_params({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
    streams, tables = pk32.pack_sdf32(d)
    o, dd, pl, near, far = make_rays(nrays, seed=1, spread=0.1)
    near, far = near.reshape(-1), far.reshape(-1)
    t = (near[:, None] + (far - near)[:, None] * np.linspace(0, 1, nper, dtype=np.float32)[None]).astype(np.float32)
    # the kernel forms the point in float32
    pts32 = (o[:, None, :] + dd[:, None, :] * t[:, :, None]).reshape(-1, 3).astype(np.float32)
    p64 = orc.params_from_state(st, torch.float64)
    sdf, feat, grad = orc.sdf_forward_grad_analytic(p64, torch.from_numpy(pts32[:ncheck].astype(np.float64)))
    out = os.path.join(ROOT, "profiles", "ubench", "bin", sys.argv[4] if len(sys.argv) > 4 else "sdf32_case.bin")
    with open(out, "wb") as f:
        hdr = np.array([nrays, nper, ncheck, streams.numel(), tables.numel()], dtype=np.int64)
        f.write(hdr.tobytes())
        f.write(streams.numpy().tobytes()); f.write(tables.numpy().astype(np.float32).tobytes())
        f.write(o.astype(np.float32).tobytes()); f.write(dd.astype(np.float32).tobytes()); f.write(t.tobytes())
        f.write(sdf.numpy().astype(np.float64).reshape(-1).tobytes())
        f.write(grad.numpy().astype(np.float64).reshape(-1).tobytes())
        f.write(feat.numpy().astype(np.float64).reshape(-1).tobytes())
    print("wrote", out, os.path.getsize(out) / 1e6, "MB")

if __name__ == "__main__":
    main()
