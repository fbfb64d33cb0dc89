#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r05; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -E "^E  |^tests/|Error|FAILED|passed|failed|skipped" | cut -c1-600 > $OUT/run5_tests.log
cat $OUT/run5_tests.log
