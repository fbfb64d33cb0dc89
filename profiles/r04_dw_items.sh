#!/bin/bash
# split-K item count of the weight-gradient launch at small batches: ms per call (dw_kernel + dw_reduce_kernel) per item count
TAG=${1:-dwitems}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r04/$TAG
mkdir -p $OUT
cd $R
for rays in 64 128 256 512; do
  timeout 200 python profiles/dw_bench.py $rays 64,128,192,256,384,512,768 2>&1 | grep "^rays" >> $OUT/dw_items.log
done
cat $OUT/dw_items.log
