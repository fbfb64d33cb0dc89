#!/bin/bash
# round 5, run 14: 16-bit hand-offs incl. the reflectance net - parity at 1 024 rays, fused-step tests, A/B of the step
mkdir -p gpurun_out/r05
O=gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_train1024.py tests/test_gpu_train_fused.py -q 2>&1 | tail -8 | tee $O/run14_tests.log
for i in 1 2; do
  NRH_DW_HALF=0 timeout 200 python profiles/train_bench.py 1024 40 graph 2>&1 | tail -1 | cut -c1-120
  NRH_DW_HALF=1 timeout 200 python profiles/train_bench.py 1024 40 graph 2>&1 | tail -1 | cut -c1-120
done | tee $O/train_half_ab2.log
