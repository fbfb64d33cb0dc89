#!/bin/bash
# round 5, run 15: kernel-level tests of the 16-bit hand-offs + per-kernel split of the 1 024-ray step with them
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_half.py -q 2>&1 | tail -60 | tee $O/run15_tests.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o train -- \
  python $R/profiles/train_bench.py 1024 10 graph > $O/prof_train.log 2>&1
f=$(find $O/prof_train -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && python $R/profiles/step_breakdown.py $f detail > $O/step_breakdown_half.txt 2>&1
rm -rf $O/prof_train
head -16 $O/step_breakdown_half.txt
