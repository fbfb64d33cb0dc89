#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r05; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_train_fused.py -q --tb=short -x -k "dw_gemm" 2>&1 | grep -E "^E  |^tests/|Error|FAILED|passed|failed|skipped" | cut -c1-500 > $OUT/run9_dw.log
cat $OUT/run9_dw.log
timeout 1500 python -m pytest tests/test_gpu_train_fused.py tests/test_gpu_train1024.py tests/test_gpu_split.py tests/test_gpu_counts.py -q --tb=short 2>&1 | grep -E "^E  |^tests/|Error|FAILED|passed|failed|skipped" | cut -c1-500 > $OUT/run9_tests.log
cat $OUT/run9_tests.log
V=nrhints_amd/lib/variants
for i in 1 2 3; do
  for v in default rowsdw; do
    lib=""; [ $v != default ] && lib=$PWD/$V/libnrh_$v.so
    echo "== $v $i" >> $OUT/train_layout_ab4.log
    NRHINTS_HIP_LIB=$lib timeout 300 python profiles/train_bench.py 1024 40 graph 2>/dev/null | tail -1 | cut -c1-140 >> $OUT/train_layout_ab4.log
  done
done
cat $OUT/train_layout_ab4.log
