#!/bin/bash
# per-kernel times of one graphed 1024-ray step: default build (all training arrays tiled) against NRH_TILE_DW=0 (only sigma' / coup tiled)
R=${GRAFT_REPO_ROOT:-.}
OUT=$R/gpurun_out/r05; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in default rowsdw; do
  lib=""; [ $v != default ] && lib=$R/nrhints_amd/lib/variants/libnrh_$v.so
  NRHINTS_HIP_LIB=$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$v -o train -- python $R/profiles/train_bench.py 1024 10 graph > $OUT/prof_$v.log 2>&1
  f=$(find $OUT/prof_$v -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python $R/profiles/step_breakdown.py $f detail > $OUT/step_breakdown_$v.txt 2>&1
  rm -rf $OUT/prof_$v
  echo "== $v"; head -12 $OUT/step_breakdown_$v.txt
done
