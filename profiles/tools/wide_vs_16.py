#!/usr/bin/env python3
"""Render the same rays with the SDF network on the wide kernels and on the 16-point kernels; print where they differ."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nrhints_amd as na
from nrhints_amd import ops
from nrhints_amd.synthetic import make_rays, perturb_state
from tests.conftest import load_npz

def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "a"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    st = load_npz("scene_a_state.npz")
    if tag == "b": st = perturb_state(st)
    m = na.NeuSHintRenderer(na.NeuSModelConfig(), precision="f16x3")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
    m = m.cuda().eval()
    o, d, pl, near, far = make_rays(n, seed=17, spread=0.15)
    cu = lambda a: torch.from_numpy(a).float().contiguous().cuda()
    rb = na.RayBundle(origins=cu(o), directions=cu(d), pl_positions=cu(pl), nears=cu(near), fars=cu(far))
    bg = torch.ones(1, 3, device="cuda")
    outs = []
    with torch.no_grad():
        for wide in (True, False, True):
            m.wide_kernels = wide
            outs.append(m(rb, is_training=False, background_rgb=bg))
    a, b, a2 = outs
    print("wide run-to-run identical:", torch.equal(a.rgb, a2.rgb), torch.equal(a.weights, a2.weights))
    for name in ("rgb", "depth", "visibilities", "weights", "analytic_normals"):
        x, y = getattr(a, name), getattr(b, name)
        dd = (x - y).abs().reshape(n, -1).max(dim=1).values
        worst = torch.topk(dd, 5)
        print(f"{name:18s} max {float(dd.max()):.3e} mean {float(dd.mean()):.3e}  worst rays {worst.indices.tolist()} {['%.2e' % v for v in worst.values.tolist()]}")
    # the SDF kernels alone on the mid points of the worst ray
    pk = m.packed_params(torch.device("cuda", 0))
    g = torch.Generator().manual_seed(1)
    pts = ((torch.rand(200000, 3, generator=g) * 2 - 1) * 0.95).cuda()
    z = torch.zeros_like(pts); t = torch.zeros(pts.shape[0], device="cuda")
    for mode in (0, 1, 2):
        s32 = ops.sdf_eval_wide(mode, pk["sdf_w32"], pk["sdf_tab32"], pts, z, t, 1)
        s16 = ops.sdf_eval(mode, pk["sdf_w"], pk["sdf_b"], pk["sdf_head"], pts, z, t, 1)
        msg = f"mode {mode}: sdf max diff {float((s32[0]-s16[0]).abs().max()):.3e}"
        if mode >= 1: msg += f" grad max diff {float((s32[1]-s16[1]).abs().max()):.3e} at {int((s32[1]-s16[1]).abs().max(dim=1).values.argmax())}"
        if mode == 2: msg += f" feat max diff {float((s32[2]-s16[2]).abs().max()):.3e}"
        print(msg)

if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "det"):
    main()


def determinism():
    """Which points differ between two runs of the wide kernel (mode 1)?"""
    st = load_npz("scene_a_state.npz")
    m = na.NeuSHintRenderer(na.NeuSModelConfig(), precision="f16x3")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
    m = m.cuda().eval()
    pk = m.packed_params(torch.device("cuda", 0))
    g = torch.Generator().manual_seed(1)
    npts = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
    pts = ((torch.rand(npts, 3, generator=g) * 2 - 1) * 0.95).cuda()
    z = torch.zeros_like(pts); t = torch.zeros(pts.shape[0], device="cuda")
    ref = ops.sdf_eval(1, pk["sdf_w"], pk["sdf_b"], pk["sdf_head"], pts, z, t, 1)[1]
    for rep in range(3):
        a = ops.sdf_eval_wide(1, pk["sdf_w32"], pk["sdf_tab32"], pts, z, t, 1)[1]
        bad = ((a - ref).abs().max(dim=1).values > 2e-3).nonzero().reshape(-1).cpu().numpy()
        print(f"run {rep}: {len(bad)} bad points of {npts}")
        if len(bad):
            grp = bad // 128; wave = (bad % 128) // 32; j = bad % 32
            print("   groups (first 20):", grp[:20].tolist())
            print("   pass index (group // 256):", np.unique(grp // 256, return_counts=True))
            print("   wave:", np.unique(wave, return_counts=True), " point-in-tile histogram:", np.bincount(j, minlength=32).tolist())
            print("   bad per group:", np.unique(np.unique(grp, return_counts=True)[1], return_counts=True))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "det":
    determinism()
