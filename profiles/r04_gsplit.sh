#!/bin/bash
# split sdf + gradient kernel: parity tests, per-pass latency against the wide kernel, step-level threshold A/B
TAG=${1:-gsplit}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r04/$TAG
mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests/test_gpu_split.py -q -x -k "gradient_kernel or small_batch" 2>&1 | tail -6
python - > $OUT/grad_split_bench.log 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, ".")
import nrhints_amd as na
from nrhints_amd import ops
from nrhints_amd.synthetic import make_rays
def timeit(fn, reps=30):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
m = na.NeuSHintRenderer(precision="f16x3").cuda().eval()
p = m.packed_params(torch.device("cuda", 0))
print("rays x 128   points   wide<1>   16pt<1>   split     (us per pass)")
for nrays in (32, 64, 96, 128, 192, 256):
    o, d, pl, near, far = (torch.from_numpy(a).cuda() for a in make_rays(nrays, seed=3, spread=0.1))
    t = (near + (far - near) * torch.linspace(0, 1, 128, device="cuda")[None]).contiguous()
    a16 = (p["sdf_w"], p["sdf_b"], p["sdf_head"], o, d, t, 128)
    r = [timeit(lambda: ops.sdf_eval_wide(1, p["sdf_w32"], p["sdf_tab32"], o, d, t, 128)), timeit(lambda: ops.sdf_eval(1, *a16)),
         timeit(lambda: ops.sdf_grad_split(*a16))]
    print(f"{nrays:>7d}     {nrays * 128:>7d} " + " ".join(f"{x:9.1f}" for x in r), flush=True)
PY
cat $OUT/grad_split_bench.log
for rep in 1 2; do
for b in 64 128; do
for m in 16384 0; do
  echo "== batch $b NRH_SPLIT_GRAD_MAX_PTS=$m" >> $OUT/gsplit_ab.log
  NRH_SPLIT_GRAD_MAX_PTS=$m timeout 200 python profiles/train_bench.py $b 40 graph 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['loss_last3'])" >> $OUT/gsplit_ab.log
done; done; done
cat $OUT/gsplit_ab.log
