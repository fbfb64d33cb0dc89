#!/bin/bash
# round 5, run 13: t as fp16 only for layers 1..6 (NRH_T16_ONLY=1) - parity at 1 024 rays and A/B
mkdir -p gpurun_out/r05
O=gpurun_out/r05
NRH_T16_ONLY=1 timeout 600 python -m pytest tests/test_gpu_train1024.py -q 2>&1 | tail -12 | tee $O/run13_t16only_tests.log
timeout 600 python -m pytest tests/test_gpu_train_fused.py -x -q 2>&1 | tail -5 | tee $O/run13_tests.log
for i in 1 2; do
  NRH_T16_ONLY=0 timeout 200 python profiles/train_bench.py 1024 40 graph 2>&1 | tail -1 | cut -c1-120
  NRH_T16_ONLY=1 timeout 200 python profiles/train_bench.py 1024 40 graph 2>&1 | tail -1 | cut -c1-120
done | tee $O/train_t16only_ab.log
