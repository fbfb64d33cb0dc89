#!/bin/bash
# tile-native `coup` (default) against the row-major form (variant couprow): graphed 1024-ray step, interleaved; sweep kernel times
TAG=${1:-coup}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r04/$TAG
mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests/test_gpu_split.py tests/test_gpu_train_fused.py -q -x -k "small_batch or fused_step_equals or gradients_vs" 2>&1 | tail -3
for rep in 1 2 3; do
for v in base couprow; do
  if [ $v = base ]; then unset NRHINTS_HIP_LIB; else export NRHINTS_HIP_LIB=$R/nrhints_amd/lib/variants/libnrh_$v.so; fi
  echo "== $v" >> $OUT/coup_ab.log
  timeout 200 python profiles/train_bench.py 1024 40 graph 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['loss_last3'])" >> $OUT/coup_ab.log
done; done
unset NRHINTS_HIP_LIB
cat $OUT/coup_ab.log
timeout 400 bash profiles/prof_train.sh r04coup 1024 graph > /dev/null 2>&1
f=$(find gpurun_out/prof_train_r04coup -name '*kernel_trace.csv' | head -1)
python profiles/step_breakdown.py $f 2>&1 | head -8 | cut -c1-100
rm -rf gpurun_out/prof_train_r04coup
