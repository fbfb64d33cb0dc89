#!/bin/bash
# kernel-level A/B of the training forward: base and the ablation variants (profiles/train_fwd_bench.py), plus the parity test
TAG=${1:-fwd1}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r04/$TAG
mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests/test_gpu_wide.py -x -q -k "training_forward" > $OUT/pytest_wide_train.log 2>&1; echo "rc=$?" >> $OUT/pytest_wide_train.log
for v in base $VARIANTS; do
  if [ $v = base ]; then unset NRHINTS_HIP_LIB; else export NRHINTS_HIP_LIB=$R/nrhints_amd/lib/variants/libnrh_$v.so; fi
  timeout 120 python profiles/train_fwd_bench.py 2>/dev/null | tail -1 >> $OUT/fwd_bench.log
done
unset NRHINTS_HIP_LIB
timeout 200 python profiles/train_bench.py 1024 30 graph 2>/dev/null | tail -1 >> $OUT/fwd_bench.log
tail -3 $OUT/pytest_wide_train.log; cat $OUT/fwd_bench.log
