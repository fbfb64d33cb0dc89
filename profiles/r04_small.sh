#!/bin/bash
# small-batch training step: graphed step time at 64 / 128 / 256 rays + a per-kernel breakdown of one step
# usage: bash profiles/r04_small.sh <tag> ["batches"]
TAG=${1:-small1}; BATCHES=${2:-"64 128"}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r04/$TAG
mkdir -p $OUT
cd $R
for b in 64 128 256 512 1024; do
  timeout 200 python profiles/train_bench.py $b 40 graph 2>/dev/null | tail -1 >> $OUT/train_bench_modes.log
done
cat $OUT/train_bench_modes.log
for b in $BATCHES; do
  timeout 400 bash profiles/prof_train.sh r04small_$b $b graph > /dev/null 2>&1
  f=$(find gpurun_out/prof_train_r04small_$b -name '*kernel_trace.csv' | head -1)
  python profiles/step_breakdown.py $f detail > $OUT/step_breakdown_$b.txt 2>&1
  head -45 $OUT/step_breakdown_$b.txt
  rm -rf gpurun_out/prof_train_r04small_$b
done
