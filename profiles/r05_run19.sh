#!/bin/bash
# round 5, run 19: work items per CU of nrh_dw_gemm with the half-operand path (the float32 kernel was flat over 1..4, profiles/r03/dw_bench_items.log)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
for rep in 1 2; do for k in 3 2 1; do
  echo -n "items per CU $k: "; NRH_DW_ITEMS_PER_CU=$k timeout 200 python profiles/train_bench.py 1024 40 graph 2>&1 | tail -1 | cut -c1-120
done; done | tee $O/dw_items_half_ab.log
