#!/usr/bin/env python3
"""Where a wave's cycles go inside the SDF kernel: needs a library built with -DNRH_TIMELINE=1
(make -C nrhints_amd/csrc variant NAME=timeline DEFS=-DNRH_TIMELINE=1; NRHINTS_HIP_LIB=.../libnrh_timeline.so).

Every wave stamps s_memtime at four points of each weight chunk (after issuing the next chunk's LDS-DMA and the
epilogue's loads | after the K loop's MFMAs | after the epilogue | after the chunk barrier) and accumulates the
differences; this prints the per-chunk averages per kernel mode and the spread over the 8 waves of a workgroup.
The stamps cost a few % themselves (scalar memory reads + lgkmcnt waits)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nrhints_amd as na
from nrhints_amd import ops, _lib
from nrhints_amd.synthetic import make_rays, perturb_state


def main():
    nrays = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    precs = sys.argv[2:] or ["f16x3", "f32"]
    lib = _lib.load()
    grid = lib.nrh_mlp_grid()
    base = na.NeuSHintRenderer()
    st = perturb_state({k: v.detach().numpy().copy() for k, v in base.state_dict().items()})
    o, d, pl, near, far = (torch.from_numpy(a).cuda() for a in make_rays(nrays, seed=1, spread=0.1))
    t = (near + (far - near) * torch.linspace(0, 1, 128, device="cuda")[None]).contiguous()
    for prec in precs:
        m = na.NeuSHintRenderer(precision=prec)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
        p = m.cuda().packed_params(torch.device("cuda", 0))
        scratch = ops._scratch(o.device)
        for mode in (0, 1, 2):
            for _ in range(2):
                ops.sdf_eval(mode, p["sdf_w"], p["sdf_b"], p["sdf_head"], o, d, t, 128, scratch=scratch)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.sdf_eval(mode, p["sdf_w"], p["sdf_b"], p["sdf_head"], o, d, t, 128, scratch=scratch)
            b.record(); torch.cuda.synchronize()
            buf = (ctypes.c_ulonglong * (grid * 64))()
            _lib.check(lib.nrh_debug_timeline_read(buf, grid * 64), "nrh_debug_timeline_read")
            tl = np.frombuffer(buf, dtype=np.uint64).reshape(grid, 8, 8).astype(np.float64)
            chunks = tl[..., 4]
            per = tl[..., :4] / chunks[..., None]                       # cycles per chunk, [wg, wave, phase]
            mean = per.mean(axis=(0, 1))
            spread = per.sum(-1).mean(0)                                # per wave slot, total cycles per chunk
            tot = mean.sum()
            print(f"{prec:6s} mode {mode}: {a.elapsed_time(b):7.2f} ms  chunks/wave {chunks.mean():8.0f}  cycles/chunk {tot:7.0f} = "
                  f"issue {mean[0]:6.0f} ({mean[0]/tot:4.0%}) | K loop {mean[1]:6.0f} ({mean[1]/tot:4.0%}) | epilogue {mean[2]:6.0f} "
                  f"({mean[2]/tot:4.0%}) | barrier {mean[3]:6.0f} ({mean[3]/tot:4.0%})   per-wave totals {np.round(spread).astype(int).tolist()}",
                  flush=True)


if __name__ == "__main__":
    main()
