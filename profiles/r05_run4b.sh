#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r05; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_counts.py "tests/test_gpu_train_fused.py::test_fused_step_off_default_branches" -q --tb=short 2>&1 | grep -E "^E  |^tests/|Error|FAILED|passed|failed" | cut -c1-700 > $OUT/run4b_tests.log
timeout 600 python -m pytest "tests/test_gpu_fullsize.py::test_bench_two_ranks_rehearsal_on_one_gpu" -q --tb=short -x 2>&1 | tail -40 | cut -c1-1500 > $OUT/run4b_rehearsal.log
cat $OUT/run4b_tests.log $OUT/run4b_rehearsal.log
