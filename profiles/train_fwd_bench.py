#!/usr/bin/env python3
"""Kernel-level timing of the SDF training forward (measurement aid): the wide kernel (nrh_sdf_train_forward_wide, sdf32_kernel<4>)
against the 16-point one (nrh_sdf_train_forward, sdf_kernel<3,1>) and the evaluation kernel of the same work without saves
(nrh_sdf_eval_wide mode 2) on 131 072 points (a 1 024-ray batch), HIP events over 20 launches each.
usage: python profiles/train_fwd_bench.py [npts]      (NRHINTS_HIP_LIB selects a variant library)"""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nrhints_amd as na
from nrhints_amd import ops, packing as pk

def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

def main():
    npts = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    torch.manual_seed(0)
    model = na.NeuSHintRenderer(precision="f16x3").cuda().eval()
    packed = model.packed_params(torch.device("cuda", 0))
    pts = ((torch.rand(npts, 3) * 2 - 1) * 0.95).cuda()
    zeros, t = torch.zeros_like(pts), torch.zeros(npts, device="cuda")
    f32 = dict(dtype=torch.float32, device="cuda")
    sdf, grad, feat = torch.empty(npts, 1, **f32), torch.empty(npts, 3, **f32), torch.empty(npts, 256, **f32)
    sv = [torch.empty(8, npts, 256, **f32) for _ in range(3)] + [torch.zeros(npts, 128, **f32)]
    scratch = ops._scratch(pts.device)
    lib, P = na._lib.load(), na._lib.ptr
    st = na._lib.stream_handle()
    w32, tab = packed["sdf_w32"], packed["sdf_tab32"]
    def wide():
        na._lib.check(lib.nrh_sdf_train_forward_wide(P(w32, w32.dtype), P(tab), P(pts), P(zeros), P(t), 1, 1, npts, P(sdf), P(grad), P(feat),
                                                     P(sv[0]), P(sv[1]), P(sv[2]), P(sv[3]), P(scratch), st), "wide")
    w16 = packed["sdf_w"]
    def k16():
        na._lib.check(lib.nrh_sdf_train_forward(1, P(w16, w16.dtype), P(packed["sdf_b"]), P(packed["sdf_head"]), P(pts), P(zeros), P(t), 1, 1, npts,
                                                P(sdf), P(grad), P(feat), P(sv[0]), P(sv[1]), P(sv[2]), P(sv[3]), st), "16pt")
    def ev():
        ops.sdf_eval_wide(2, w32, tab, pts, zeros, t, 1, scratch=scratch)
    out = {"npts": npts, "lib": os.path.basename(os.environ.get("NRHINTS_HIP_LIB", "libnrhints_hip.so")),
           "wide_train_ms": round(timed(wide), 4), "k16_train_ms": round(timed(k16), 4), "wide_eval_mode2_ms": round(timed(ev), 4)}
    print(json.dumps(out))

if __name__ == "__main__":
    main()
