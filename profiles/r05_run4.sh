#!/bin/bash
# round 5, GPU call 4: sample-count variants, the fused step's new branches, regression of the suites the edits touch, eval A/B
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r05; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_counts.py tests/test_gpu_train_fused.py tests/test_gpu_train1024.py -q 2>&1 | tail -40 > $OUT/run4_tests_a.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -15 > $OUT/run4_tests_b.log
for i in 1 2; do
  timeout 300 python bench.py --steps 4 --warmup 2 --cpu-rays 0 --no-train --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])" >> $OUT/eval_run4.log
done
cat $OUT/run4_tests_a.log $OUT/run4_tests_b.log $OUT/eval_run4.log
