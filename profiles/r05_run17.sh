#!/bin/bash
# round 5, run 17: 16-bit hand-offs on the 4-wave builds (65..128 rays): kernel tests, small-batch fused tests, A/B at 128 / 64 rays
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_half.py tests/test_gpu_split.py tests/test_gpu_train_fused.py -q 2>&1 | grep -v "tensor(\|^E    \s*+" | grep "^E \|passed\|failed\|py:[0-9]*: " | cut -c1-300 | tee $O/run17_tests.log
for b in 128 64; do for h in 0 1; do
  echo -n "rays $b NRH_DW_HALF=$h: "; NRH_DW_HALF=$h timeout 200 python profiles/train_bench.py $b 60 graph 2>&1 | tail -1 | cut -c1-130
done; done | tee $O/train_half_small_ab.log
