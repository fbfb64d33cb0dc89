#!/bin/bash
# round 5, run 16: coup as fp16 with the 16-bit hand-offs - kernel tests, parity at 1 024 rays, A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_half.py tests/test_gpu_train1024.py tests/test_gpu_train_fused.py -q 2>&1 | grep -v "tensor(\|^E    \s*+" | tail -25 | cut -c1-300 | tee $O/run16_tests.log
for i in 1 2; do
  NRH_COUP16=0 timeout 200 python profiles/train_bench.py 1024 40 graph 2>&1 | tail -1 | cut -c1-120
  NRH_COUP16=1 timeout 200 python profiles/train_bench.py 1024 40 graph 2>&1 | tail -1 | cut -c1-120
done | tee $O/train_coup16_ab.log
