#!/usr/bin/env python3
"""Per-launch ns per point of the wide kernels INSIDE an 800x800 evaluation frame, from a rocprofv3 --kernel-trace CSV of
`python bench.py --steps K --warmup 1 --cpu-rays 0 --no-train --no-secondary` (is the 16-sample pass really
30 % faster per point than the 64-sample pass of the same kernel?).

A frame is 5 chunks (4 x 131 072 rays + 115 712); per chunk nrh_render_forward launches, in order, sdf32<0> eight times - primary
ray: 64 samples per ray, then 3 x 16; shadow ray: 64, then 3 x 16 - sdf32<2> once (128 per ray), sdf32<1> once (128 per ray) and
color32 once: the position of a launch in that sequence says how many points it had.

    python profiles/frame_trace_analyze.py <..._kernel_trace.csv> [frames to skip = 1]"""
import csv
import sys
from collections import defaultdict

import numpy as np


def main():
    path = sys.argv[1]
    skip_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    kinds = {"sdf32_kernel<0>": "sdf0", "sdf32_kernel<1>": "sdf1", "sdf32_kernel<2>": "sdf2", "color32_kernel": "col"}
    seq = defaultdict(list)
    for s, e, name in rows:
        for key, k in kinds.items():
            if key in name and "nrh32t" not in name:
                seq[k].append((s, e))
    chunk_rays = [131072] * 4 + [115712]
    out = defaultdict(list)
    t_first = rows[0][0]
    for k, per_chunk in (("sdf0", 8), ("sdf1", 1), ("sdf2", 1), ("col", 1)):
        launches = seq[k]
        per_frame = per_chunk * 5
        for i, (s, e) in enumerate(launches):
            frame, j = divmod(i, per_frame)
            if frame < skip_frames:
                continue
            chunk, pos = divmod(j, per_chunk)
            per_ray = (64 if pos % 4 == 0 else 16) if k == "sdf0" else 128
            npts = chunk_rays[chunk] * per_ray
            label = (k, per_ray, "full chunk" if chunk < 4 else "last chunk (115 712 rays)",
                     ("primary" if pos < 4 else "shadow") if k == "sdf0" else "")
            out[label].append((e - s) / npts)
    print(f"{'kernel':6s} {'per ray':>7s} {'chunk':28s} {'ray':8s} {'launches':>8s} {'ns/pt mean':>10s} {'min':>7s} {'max':>7s}")
    for label in sorted(out):
        v = np.array(out[label])
        print(f"{label[0]:6s} {label[1]:7d} {label[2]:28s} {label[3]:8s} {len(v):8d} {v.mean():10.3f} {v.min():7.3f} {v.max():7.3f}")
    # and the plain sequence of one frame, for drift along the frame
    k = "sdf0"
    per_frame = 40
    fr = seq[k][skip_frames * per_frame:(skip_frames + 1) * per_frame]
    if fr:
        print("\none frame's sdf32<0> launches in order (ms since the frame's first, ns/pt):")
        for j, (s, e) in enumerate(fr):
            chunk, pos = divmod(j, 8)
            per_ray = 64 if pos % 4 == 0 else 16
            print(f"  chunk {chunk} pos {pos} ({per_ray:2d}/ray): t = {(s - fr[0][0]) / 1e6:8.2f} ms  {(e - s) / (chunk_rays[chunk] * per_ray):.3f} ns/pt  ({(e - s) / 1e6:.3f} ms)")


if __name__ == "__main__":
    main()
