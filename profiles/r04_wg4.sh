#!/bin/bash
# 16-point kernels with 4 waves per workgroup (one wave per SIMD at small batches) against the default 8: graphed training step
TAG=${1:-wg4}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r04/$TAG
mkdir -p $OUT
cd $R
for rep in 1 2; do
for b in 64 128 256; do
for v in base wg4; do
  if [ $v = base ]; then unset NRHINTS_HIP_LIB; else export NRHINTS_HIP_LIB=$R/nrhints_amd/lib/variants/libnrh_$v.so; fi
  echo "== batch $b $v" >> $OUT/wg4_ab.log
  timeout 200 python profiles/train_bench.py $b 40 graph 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['loss_last3'])" >> $OUT/wg4_ab.log
done; done; done
unset NRHINTS_HIP_LIB
cat $OUT/wg4_ab.log
export NRHINTS_HIP_LIB=$R/nrhints_amd/lib/variants/libnrh_wg4.so
timeout 400 bash profiles/prof_train.sh r04wg4 64 graph > /dev/null 2>&1
f=$(find gpurun_out/prof_train_r04wg4 -name '*kernel_trace.csv' | head -1)
python profiles/step_breakdown.py $f detail 2>&1 | head -16
rm -rf gpurun_out/prof_train_r04wg4
