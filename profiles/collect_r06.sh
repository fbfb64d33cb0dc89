#!/bin/bash
# Round-6 evidence in ONE gpurun call (GPU box, repo root):  bash profiles/collect_r06.sh <tag>
#   1. the default bench.py line (incl. train, train_small, train_camopt, register_view, cpu_baseline, secondary x3) with rocm-smi beside it
#   2. rocprofv3 --kernel-trace --stats of the bench command (evaluation render only)
#   3. rocprofv3 --kernel-trace --stats of the graphed 1024-ray training step + the per-kernel split of one step
#   4. the PMC passes of profiles/pmc_run.sh (evaluation) and profiles/pmc_train.sh (training step): separate --pmc runs
# Everything lands in gpurun_out/r06/<tag>/; what is to be judged is copied into profiles/r06/.
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r06/$TAG
mkdir -p $OUT
cd $R
( while true; do
    echo "t=$(date +%s.%N | cut -c1-14) $(rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E 'sclk|Package Power|Sensor junction' | sed -e 's/GPU\[0\]\s*: //' | tr '\n' '|')"
    sleep 1
  done ) > $OUT/smi_during_bench.log 2>&1 &
SMI=$!
timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
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
cd $R
for b in 64 128 512 1024; do
  timeout 200 python profiles/train_bench.py $b 40 graph 2>/dev/null | tail -1 >> $OUT/train_bench_modes.log            # float32 hand-offs (the default)
  timeout 200 python profiles/train_bench.py $b 40 graph half 2>/dev/null | tail -1 >> $OUT/train_bench_modes.log       # the fp16 dW hand-offs, labelled
done
# the per-launch trace analysis of an evaluation frame (is the kernel's cost per point flat across launch sizes?)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $R/bench.py --steps 3 --warmup 1 --cpu-rays 0 --no-train --no-secondary > /dev/null 2>&1
f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && python $R/profiles/frame_trace_analyze.py $f 1 > $OUT/frame_trace_per_launch.txt 2>&1
rm -rf $OUT/trace
cd $R
bash profiles/pmc_run.sh r06_$TAG f16x3 > $OUT/pmc.log 2>&1
mkdir -p $OUT/pmc_f16x3 && cp gpurun_out/pmc_r06_$TAG/summary.txt $OUT/pmc_f16x3/summary.txt 2>/dev/null
rm -rf gpurun_out/pmc_r06_$TAG
bash profiles/pmc_train.sh r06_$TAG > $OUT/pmc_train.log 2>&1
cp gpurun_out/pmc_train_r06_$TAG/summary.txt $OUT/pmc_train_summary.txt 2>/dev/null
rm -rf gpurun_out/pmc_train_r06_$TAG
echo done; tail -c 3000 $OUT/bench.json
