#!/bin/bash
# round 5, GPU call 2: gradient parity at 1 024 rays with the scaled adjoint chains (adj_scale), then the training / parity suites
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r05; mkdir -p $OUT
python profiles/train1024_diag.py > $OUT/train1024_diag_scaled.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_train1024.py tests/test_gpu_split.py tests/test_gpu_train_fused.py tests/test_gpu_parity.py -q 2>&1 | tail -30 > $OUT/run2_tests.log
for i in 1 2; do python profiles/train_bench.py 1024 40 graph 2>/dev/null | tail -1 >> $OUT/train_scaled.log; done
grep -v "^   ratio" $OUT/train1024_diag_scaled.log; cat $OUT/run2_tests.log $OUT/train_scaled.log
