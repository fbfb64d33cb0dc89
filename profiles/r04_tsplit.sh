#!/bin/bash
# channel-split training kernels against the 4-wave builds: graphed step A/B (interleaved, twice) + one step's kernels
TAG=${1:-tsplit}; BATCHES=${2:-"64 128"}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r04/$TAG
mkdir -p $OUT
cd $R
for rep in 1 2; do
for b in $BATCHES; do
for v in 1 0; do
  echo "== batch $b NRH_SPLIT_TRAIN=$v" >> $OUT/tsplit_ab.log
  NRH_SPLIT_TRAIN=$v timeout 200 python profiles/train_bench.py $b 40 graph 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['loss_last3'])" >> $OUT/tsplit_ab.log
done; done; done
cat $OUT/tsplit_ab.log
for b in $BATCHES; do
  timeout 400 bash profiles/prof_train.sh r04ts_$b $b graph > /dev/null 2>&1
  f=$(find gpurun_out/prof_train_r04ts_$b -name '*kernel_trace.csv' | head -1)
  python profiles/step_breakdown.py $f detail > $OUT/step_breakdown_$b.txt 2>&1
  head -14 $OUT/step_breakdown_$b.txt | cut -c1-120
  rm -rf gpurun_out/prof_train_r04ts_$b
done
