#!/bin/bash
# Experiment build of libnrhints_hip.so with its own generated schedules:  build_gen_variant.sh NAME "ENV=.. ENV2=.." "-DDEF=.."
#   -> nrhints_amd/lib/variants/libnrh_NAME.so (git-ignored, travels with gpurun; select it with NRHINTS_HIP_LIB).
# Builds from a shadow copy of csrc/ with its own gen32/ (the generator reads its knobs from the environment); the api TU is
# reused from the main build unless defines are given.  Prints the wide kernels' register counts and the ISA check.
set -e
NAME=$1; GENV=$2; DEFS=$3
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CS=$ROOT/nrhints_amd/csrc
W=/tmp/nrh_var_$NAME; rm -rf $W; mkdir -p $W/csrc $ROOT/nrhints_amd/lib/variants
cp $CS/*.h $CS/*.hip $CS/*.py $W/csrc/
( cd $W/csrc && env $GENV python3 gen_mlp32.py gen32 >/dev/null )
FLAGS="-O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -fPIC"
sed -i "s#\"../../include/nrhints_hip.h\"#\"$ROOT/include/nrhints_hip.h\"#" $W/csrc/nrh_api.hip
API=$ROOT/nrhints_amd/lib/obj/nrh_api.o
if [ -n "$DEFS" ]; then ( cd $W/csrc && /opt/rocm/bin/hipcc $FLAGS $DEFS -c -o api.o nrh_api.hip ) & API=$W/csrc/api.o; fi
( cd $W/csrc && /opt/rocm/bin/hipcc $FLAGS -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form $DEFS -save-temps=obj -c -o wide.o nrh_wide.hip )
wait
python3 $CS/check_wide_isa.py $W/csrc/nrh_wide-hip-amdgcn-amd-amdhsa-gfx950.s
grep -E "^\s+\.(name|vgpr_count|vgpr_spill_count):" $W/csrc/nrh_wide-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - | sed 's/  */ /g' | grep "sdf32_kernel" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/nrhints_amd/lib/variants/libnrh_$NAME.so $API $W/csrc/wide.o $ROOT/nrhints_amd/lib/obj/nrh_wide1.o $ROOT/nrhints_amd/lib/obj/nrh_small.o   # (the 4-wave training kernels: main build's unit)
