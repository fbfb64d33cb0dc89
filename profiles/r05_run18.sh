#!/bin/bash
# round 5, run 18: why bench.py's train_small (64 rays) and train_camopt legs slowed down with the 16-bit hand-offs on
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
for h in 1 0; do
  echo "NRH_DW_HALF=$h"
  NRH_DW_HALF=$h timeout 400 python bench.py --steps 1 --warmup 1 --cpu-rays 0 --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('train', d['train']['value'], 'small', d['train_small']['ms_per_step'], 'camopt', d['train_camopt']['value'], d['train_camopt']['mode'][:40], 'register_view', d['register_view']['value'])"
done | tee $O/bench_legs_half_ab.log
