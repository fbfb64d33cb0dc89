#!/bin/bash
# Round-3 evidence in ONE gpurun call (GPU box, repo root):  bash profiles/collect_r03.sh <tag>
#   1. the default bench.py line with rocm-smi (sclk / package power / junction temperature) polled beside it once per second
#   2. rocprofv3 --kernel-trace --stats of the bench command (evaluation render only)
#   3. rocprofv3 --kernel-trace --stats of the graphed 1024-ray training step + the per-category split of one step
#   4. the PMC passes of profiles/pmc_run.sh (separate --pmc runs with --kernel-trace only)
# Everything lands in gpurun_out/r03/<tag>/; copy what is to be judged into profiles/r03/.
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r03/$TAG
mkdir -p $OUT
cd $R

( while true; do
    echo "t=$(date +%s.%N | cut -c1-14) $(rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E 'sclk|Package Power|Sensor junction' | sed -e 's/GPU\[0\]\s*: //' | tr '\n' '|')"
    sleep 1
  done ) > $OUT/smi_during_bench.log 2>&1 &
SMI=$!
echo "bench_start $(date +%s.%N | cut -c1-14)" > $OUT/bench_times.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench_end $(date +%s.%N | cut -c1-14)" >> $OUT/bench_times.txt
kill $SMI

cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_eval -o eval -- \
  python $R/bench.py --steps 2 --warmup 1 --cpu-rays 0 --no-train --no-secondary > $OUT/prof_eval.log 2>&1
f=$(find $OUT/prof_eval -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/rocprof_kernel_stats_eval.csv
rm -rf $OUT/prof_eval

timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o train -- \
  python $R/profiles/train_bench.py 1024 10 graph > $OUT/prof_train.log 2>&1
f=$(find $OUT/prof_train -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/rocprof_train_stats.csv
f=$(find $OUT/prof_train -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && python $R/profiles/step_breakdown.py $f detail > $OUT/train_step_breakdown.txt 2>&1
rm -rf $OUT/prof_train
python - $OUT/rocprof_train_stats.csv > $OUT/train_stats_summary.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
nat = sum(float(r["TotalDurationNs"]) for r in rows if "at::native" in r["Name"] or "rocclr" in r["Name"])
blas = [r["Name"] for r in rows if r["Name"].startswith("Cijk") or "rocblas" in r["Name"].lower() or "Tensile" in r["Name"]]
print(f"whole run (teacher renders + capture warm-up + 13 replayed steps): {tot / 1e6:.2f} ms of kernel time in {len(rows)} distinct kernels")
print(f"rocBLAS / Tensile kernels (Cijk_*): {len(blas)} rows")
print(f"at::native + runtime copy kernels: {nat / 1e6:.3f} ms = {100 * nat / tot:.2f} % of the kernel time")
PY

cd $R
bash profiles/pmc_run.sh r03_$TAG f16x3 > $OUT/pmc.log 2>&1
cp gpurun_out/pmc_r03_$TAG/summary.txt $OUT/pmc_summary.txt 2>/dev/null
rm -rf gpurun_out/pmc_r03_$TAG
echo done
