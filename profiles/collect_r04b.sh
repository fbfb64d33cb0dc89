#!/bin/bash
# Round-4 evidence, second pass (after the small-batch kernels; the evaluation kernels' sources - and with them the counter summary
# pmc_f16x3_v13 - are unchanged):  bash profiles/collect_r04b.sh <tag>
#   bench.py line | rocprofv3 kernel stats of the 1024-ray training step | small-batch logs (step times, per-pass latencies, breakdowns)
TAG=${1:-v3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r04/$TAG
mkdir -p $OUT
cd $R
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o train -- \
  python $R/profiles/train_bench.py 1024 10 graph > $OUT/prof_train.log 2>&1
f=$(find $OUT/prof_train -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/rocprof_train_stats.csv
f=$(find $OUT/prof_train -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && python $R/profiles/step_breakdown.py $f detail > $OUT/train_step_breakdown.txt 2>&1
rm -rf $OUT/prof_train
cd $R
bash profiles/r04_small.sh $TAG "64 128" > $OUT/small.log 2>&1
timeout 300 python profiles/split_bench.py > $OUT/split_bench.log 2>&1
echo done
