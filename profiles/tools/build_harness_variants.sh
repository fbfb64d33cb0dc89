#!/bin/bash
# Build stand-alone sdf32_bench binaries for several generator settings (A/B on the GPU box in one call).
#   profiles/tools/build_harness_variants.sh "tag1|ENV1=.. ENV2=..|-DDEF.." "tag2|..|.." ...   (generator env | compiler defines)
# -> profiles/ubench/bin/sdf32_bench_<tag>  (data/ is git-ignored but travels with gpurun).  Each variant compiles from its own
# shadow copy of csrc/ (symlinked sources + its own gen32/), so the builds run in parallel and the tree's gen32/ is untouched.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CS=$ROOT/nrhints_amd/csrc
mkdir -p $ROOT/profiles/ubench/bin
for spec in "$@"; do
  IFS='|' read -r tag envs defs <<< "$spec"
  W=/tmp/nrh_variant_$tag; rm -rf $W; mkdir -p $W
  for f in $CS/*.h $CS/*.hip $CS/*.py; do ln -s $f $W/; done
  ( cd $W && env $envs python3 gen_mlp32.py $W/gen32 >/dev/null )
  sed "s#include \"nrh_sdf32.hip\"#include \"$W/nrh_sdf32.hip\"#" $ROOT/profiles/ubench/sdf32_bench.hip > $W/bench.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form $defs \
      -I $W $W/bench.hip -o $ROOT/profiles/ubench/bin/sdf32_bench_$tag &
done
wait
ls -la $ROOT/profiles/ubench/bin/ | grep sdf32_bench_
