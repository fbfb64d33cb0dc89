#!/bin/bash
# Experiment build: libnrhints_hip.so with the reflectance net's activations handed on with the UNSCALED
# residual (gen_mlp32.py NRH32_COL_UNSCALED=1, nrh_color32.hip -DNRH32_COL_UNSCALED=1) -> nrhints_amd/lib/variants/libnrh_colu.so
# (git-ignored, travels with gpurun; select it with NRHINTS_HIP_LIB).  Builds from a shadow copy of csrc/ with its own gen32/.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CS=$ROOT/nrhints_amd/csrc
W=/tmp/nrh_colu; rm -rf $W; mkdir -p $W/csrc $W/include $ROOT/nrhints_amd/lib/variants $ROOT/profiles/ubench/data
cp $CS/*.h $CS/*.hip $CS/*.py $W/csrc/; cp $ROOT/include/*.h $W/include/
mkdir -p $W/x/y; ln -s $W/include $W/x/include 2>/dev/null || true
( cd $W/csrc && NRH32_COL_UNSCALED=1 python3 gen_mlp32.py gen32 >/dev/null )
FLAGS="-O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -fPIC"
sed -i "s#\"../../include/nrhints_hip.h\"#\"$ROOT/include/nrhints_hip.h\"#" $W/csrc/nrh_api.hip
( cd $W/csrc && /opt/rocm/bin/hipcc $FLAGS -c -o api.o nrh_api.hip &
  cd $W/csrc && /opt/rocm/bin/hipcc $FLAGS -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form -DNRH32_COL_UNSCALED=1 -save-temps=obj -c -o wide.o nrh_wide.hip &
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $ROOT/profiles/ubench/denorm_path.hip -o $ROOT/profiles/ubench/data/denorm_path &
  wait )
python3 $CS/check_wide_isa.py $W/csrc/nrh_wide-hip-amdgcn-amd-amdhsa-gfx950.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/nrhints_amd/lib/variants/libnrh_colu.so $W/csrc/api.o $W/csrc/wide.o
ls -la $ROOT/nrhints_amd/lib/variants/libnrh_colu.so $ROOT/profiles/ubench/data/denorm_path
