#!/bin/bash
# threshold A/B of the split sampler kernel in the graphed training step: NRH_SPLIT_MAX_PTS in {0, 4096, 8192, 16384}
TAG=${1:-splitab}; BATCHES=${2:-"128 1024"}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r04/$TAG
mkdir -p $OUT
cd $R
for rep in 1 2; do
for b in $BATCHES; do
for m in 0 4096 8192 16384; do
  echo "== batch $b  NRH_SPLIT_MAX_PTS=$m" >> $OUT/split_ab.log
  NRH_SPLIT_MAX_PTS=$m timeout 200 python profiles/train_bench.py $b 40 graph 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $OUT/split_ab.log
done; done; done
cat $OUT/split_ab.log
for b in 128; do
  timeout 400 bash profiles/prof_train.sh r04ab_$b $b graph > /dev/null 2>&1
  f=$(find gpurun_out/prof_train_r04ab_$b -name '*kernel_trace.csv' | head -1)
  python profiles/step_breakdown.py $f detail 2>&1 | head -12
  rm -rf gpurun_out/prof_train_r04ab_$b
done
