#!/usr/bin/env python3
"""Per-category GPU time of ONE training step from a rocprofv3 kernel trace (profiles/prof_train.sh output).
usage: python profiles/step_breakdown.py gpurun_out/prof_train_<tag>/train_kernel_trace.csv"""
import collections, csv, sys

DETAIL = len(sys.argv) > 2 and sys.argv[2] == "detail"
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if any(k in r["Kernel_Name"] for k in ("sdf_kernel<3", "sdf32_kernel<4>", "sdf_train_split_kernel"))]   # the training forward marks a step


def step_start(i):
    while i > 0 and "coarse_z" not in rows[i]["Kernel_Name"]:
        i -= 1
    return i


step = rows[step_start(idx[-2]):step_start(idx[-1])]
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])


def cat(name):
    if "nrh::" in name or "nrhdw::" in name or "nrhadam::" in name or "nrh32::" in name:
        return name.split("(")[0].replace("void ", "")[:44]
    if name.startswith("Cijk"):
        return "rocBLAS GEMM " + name[5:14] + " " + name[name.index("MT"):name.index("MT") + 12]
    if DETAIL and "at::native" in name:
        import re
        f = re.findall(r"at::native::(?:\(anonymous namespace\)::)?(?:binary_internal::)?(\w+(?:Functor|Kernel|kernel|Copy\w*|Op)\w*)", name)
        return "torch: " + "/".join(dict.fromkeys(f[-3:]))[:70] if f else "torch: " + name[:60]
    for key, label in (("reduce_kernel", "torch reduce"), ("elementwise", "torch elementwise"), ("vectorized", "torch elementwise"),
                       ("Cat", "torch cat"), ("copyBuffer", "memcpy/fill"), ("fillBuffer", "memcpy/fill"),
                       ("multi_tensor", "adam/foreach"), ("index", "torch index/scatter")):
        if key in name:
            return label
    return name[:44]


agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    k = cat(r["Kernel_Name"])
    agg[k][0] += dur(r)
    agg[k][1] += 1
span = int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])
print(f"one training step: {len(step)} kernel launches, {sum(dur(r) for r in step) / 1e6:.3f} ms of kernel time, {span / 1e6:.3f} ms from the first kernel's start to the last one's end")
gaps = sorted(((int(step[i + 1]["Start_Timestamp"]) - int(step[i]["End_Timestamp"])) / 1e3, cat(step[i]["Kernel_Name"]), cat(step[i + 1]["Kernel_Name"])) for i in range(len(step) - 1))
if DETAIL:
    print("largest gaps (us, after, before):", [(round(g, 1), a[:28], b[:28]) for g, a, b in gaps[-5:]])
for k, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    if t > 5000 or DETAIL:
        print(f"{t / 1e6:8.3f} ms {n:5d}  {k}")
if len(sys.argv) > 2 and sys.argv[2] in ("sequence", "detail+sequence") or (len(sys.argv) > 3 and sys.argv[3] == "sequence"):
    t0 = int(step[0]["Start_Timestamp"])
    print("\nthe step's launches in order (us since its first kernel, duration us):")
    for r in step:
        print(f"  {(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {dur(r) / 1e3:8.1f}  {cat(r['Kernel_Name'])}")
