#!/bin/bash
# round 5, GPU call 6: layout A/B of the training arrays after the spill analysis (profiles/r05/spill_counts.log), + the touched tests
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r05; mkdir -p $OUT
V=nrhints_amd/lib/variants
for i in 1 2 3; do
  for v in default rot s1tile allrows; do
    lib=""; [ $v != default ] && lib=$PWD/$V/libnrh_$v.so
    echo "== $v $i" >> $OUT/train_layout_ab.log
    NRHINTS_HIP_LIB=$lib timeout 300 python profiles/train_bench.py 1024 40 graph 2>/dev/null | tail -1 | cut -c1-140 >> $OUT/train_layout_ab.log
  done
done
timeout 1500 python -m pytest tests/test_gpu_counts.py tests/test_gpu_train_fused.py tests/test_gpu_train1024.py tests/test_gpu_split.py -q --tb=short 2>&1 | grep -E "^E  |^tests/|Error|FAILED|passed|failed|skipped" | cut -c1-600 > $OUT/run6_tests.log
cat $OUT/train_layout_ab.log $OUT/run6_tests.log
