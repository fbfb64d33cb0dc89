#!/bin/bash
# usage: pmc_train.sh <tag>   (GPU box, repo root; writes gpurun_out/pmc_train_<tag>/)
# Counter passes over the eager fused training step (profiles/train_bench.py 1024 6 hip_nosync): HBM bytes and instruction mix per
# kernel of the step.  One rocprofv3 --pmc pass per counter group, each with --kernel-trace only.
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_train_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- python $GRAFT_REPO_ROOT/profiles/train_bench.py 1024 6 hip_nosync > $OUT/$n.log 2>&1
}
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES
run tcc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python $GRAFT_REPO_ROOT/profiles/pmc_summarize.py $OUT > $OUT/summary.txt 2>&1
