#!/bin/bash
# Repeat the stand-alone wide-kernel harness and collect the checksums (sdf, gradient, features over all 4.19 M points) of every run:
# a schedule race (a fragment or weight block read before it has landed) shows up as a run whose sums differ.
#   stress_checksums.sh OUT N variant [variant ...]      (variants: profiles/ubench/bin/sdf32_bench_<variant>)
cd $GRAFT_REPO_ROOT
OUT=$1; N=$2; shift 2
mkdir -p $(dirname $OUT)
for v in "$@"; do
  for i in $(seq $N); do
    timeout 100 profiles/ubench/bin/sdf32_bench_$v profiles/ubench/bin/sdf32_case.bin 1 2>&1 | grep "^mode" | sed -e "s/^\(mode [0-9]\).*nan \([0-9]*\)  \(sum.*\)$/$v \1 nan \2 \3/"
  done
done | sort | uniq -c > $OUT
cat $OUT
