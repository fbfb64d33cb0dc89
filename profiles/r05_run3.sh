#!/bin/bash
# round 5, GPU call 3: the whole GPU suite after the adjoint scaling + the pooled yardstick, smoke, a short bench line
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r05; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $OUT/run3_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/run3_smoke.log 2>&1
timeout 900 python bench.py --steps 3 --warmup 1 > $OUT/bench_run3.json 2> $OUT/bench_run3.err
cat $OUT/run3_tests.log; tail -4 $OUT/run3_smoke.log; cat $OUT/bench_run3.json | head -c 6000
