#!/bin/bash
# evaluation-render A/B of variant libraries: bash profiles/r04_eval_ab.sh <tag> "<variants>"   (each twice, interleaved with base)
TAG=${1:-ev1}; VARS=$2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r04/$TAG
mkdir -p $OUT
cd $R
for rep in 1 2; do
for v in base $VARS; do
  if [ $v = base ]; then unset NRHINTS_HIP_LIB; else export NRHINTS_HIP_LIB=$R/nrhints_amd/lib/variants/libnrh_$v.so; fi
  echo "== $v" >> $OUT/eval_ab.log
  timeout 200 python bench.py --steps 3 --warmup 1 --cpu-rays 0 --no-train --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])" >> $OUT/eval_ab.log
done
done
for v in $VARS; do
  export NRHINTS_HIP_LIB=$R/nrhints_amd/lib/variants/libnrh_$v.so
  timeout 400 python -m pytest tests/test_gpu_wide.py -x -q > $OUT/pytest_wide_$v.log 2>&1; echo "rc=$?" >> $OUT/pytest_wide_$v.log; tail -2 $OUT/pytest_wide_$v.log
done
cat $OUT/eval_ab.log
