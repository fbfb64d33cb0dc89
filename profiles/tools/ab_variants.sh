#!/bin/bash
# Interleaved headline A/B of library variants on one box:  ab_variants.sh OUTDIR ROUNDS name1 name2 ...   ("main" = the tree's library,
# anything else = nrhints_amd/lib/variants/libnrh_<name>.so through NRHINTS_HIP_LIB); prints rays/s, ms per frame, roofline.frac and
# the dominant kernel's launch time per run.
cd $GRAFT_REPO_ROOT
OUT=$1; R=$2; shift 2
mkdir -p $OUT
for i in $(seq $R); do
 for v in "$@"; do
  if [ $v = main ]; then unset NRHINTS_HIP_LIB; else export NRHINTS_HIP_LIB=$PWD/nrhints_amd/lib/variants/libnrh_$v.so; fi
  timeout 600 python bench.py --steps 4 --warmup 1 --cpu-rays 0 --no-train --no-secondary --no-camopt 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        j = json.loads(l); print('$v', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['avg_launch_ms'])
" >> $OUT/ab.log
 done
done
cat $OUT/ab.log
