#!/bin/bash
# channel-split sampler kernel: parity tests, per-pass latency, graphed training step at 64..1024 rays (+ breakdown at 64 / 128)
TAG=${1:-split1}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r04/$TAG
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_split.py -x -q > $OUT/pytest_split.log 2>&1; echo "rc=$?" >> $OUT/pytest_split.log; tail -15 $OUT/pytest_split.log
timeout 300 python profiles/split_bench.py > $OUT/split_bench.log 2>&1; cat $OUT/split_bench.log
bash profiles/r04_small.sh $TAG "64 128" 2>&1 | grep -v "^ *0.00"
