#!/bin/bash
# Round-4 A/B in one gpurun call: the new parity tests, the training leg with the wide training forward against the 16-point one
# and two cache-policy variants, and the per-kernel split of one graphed step.   bash profiles/r04_ab.sh <tag>
TAG=${1:-ab1}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r04/$TAG
mkdir -p $OUT
cd $R
timeout 500 python -m pytest tests/test_gpu_wide.py -x -q -k "training_forward" > $OUT/pytest_wide_train.log 2>&1; echo "rc=$?" >> $OUT/pytest_wide_train.log
timeout 600 python -m pytest tests/test_gpu_train_fused.py tests/test_gpu_parity2.py -x -q -m gpu -k "fused or hip_adam or free_scalars or train_step or graph" > $OUT/pytest_train.log 2>&1; echo "rc=$?" >> $OUT/pytest_train.log
for v in base fwd16 base fwd16; do
  if [ $v = base ]; then unset NRHINTS_HIP_LIB; else export NRHINTS_HIP_LIB=$R/nrhints_amd/lib/variants/libnrh_$v.so; fi
  echo "== $v" >> $OUT/train_ab.log
  timeout 200 python profiles/train_bench.py 1024 30 graph 2>/dev/null | tail -1 >> $OUT/train_ab.log
done
unset NRHINTS_HIP_LIB
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o train -- python $R/profiles/train_bench.py 1024 10 graph > $OUT/prof_train.log 2>&1
f=$(find $OUT/prof_train -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && python $R/profiles/step_breakdown.py $f detail > $OUT/train_step_breakdown.txt 2>&1
f=$(find $OUT/prof_train -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/rocprof_train_stats.csv
rm -rf $OUT/prof_train
tail -4 $OUT/pytest_wide_train.log; tail -4 $OUT/pytest_train.log; cat $OUT/train_ab.log; head -14 $OUT/train_step_breakdown.txt
