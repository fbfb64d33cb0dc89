#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r05; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_wide.py -q --tb=short 2>&1 | grep -E "^E  |^tests/|Error|FAILED|passed|failed|skipped" | cut -c1-600 > $OUT/run11_tests.log
cat $OUT/run11_tests.log
timeout 600 python bench.py --steps 3 --warmup 1 --cpu-rays 0 --no-train 2>$OUT/run11_bench.err | tail -1 > $OUT/run11_bench.json
python - <<'P'
import json
d=json.load(open('gpurun_out/r05/run11_bench.json'))
print('headline', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])
print('reduced', d.get('reduced'))
P
