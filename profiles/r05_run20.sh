#!/bin/bash
# round 5, run 20: per-kernel split of the 64-ray and 128-ray graphed steps
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for b in 64 128; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_small -o t -- python $R/profiles/train_bench.py $b 10 graph > $O/prof_small.log 2>&1
  f=$(find $O/prof_small -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python $R/profiles/step_breakdown.py $f detail > $O/step_breakdown_${b}rays.txt 2>&1
  rm -rf $O/prof_small
  head -22 $O/step_breakdown_${b}rays.txt | cut -c1-150
done
