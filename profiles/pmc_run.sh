#!/bin/bash
# usage: pmc_run.sh <tag> <precision>   (run on the GPU box from the repo root; writes gpurun_out/pmc_<tag>/)
# One rocprofv3 --pmc pass per counter group (SQ has 8 slots, TCC 4, FETCH_SIZE and WRITE_SIZE need a pass each).
TAG=$1; PREC=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-rays 0 --no-train --no-secondary --precision $PREC > $OUT/$n.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD
run tcc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python $GRAFT_REPO_ROOT/profiles/pmc_summarize.py $OUT > $OUT/summary.txt 2>&1
# which kernels these counters belong to (bench.py quotes the summary only while this matches its own hash of the kernel sources)
( cd $GRAFT_REPO_ROOT && python -c "import bench; print('source_hash', bench.kernel_source_hash())" ) >> $OUT/summary.txt 2>/dev/null
# ... and how many rays a launch of the dominant kernel had in these runs (bench.py scales the per-launch traffic to its own launch size)
( cd $GRAFT_REPO_ROOT && python -c "
import nrhints_amd as na
c = na.NeuSHintRenderer.whole_frame_rays
print('rays_per_launch', 640000 if c >= 640000 else 128000)" ) >> $OUT/summary.txt 2>/dev/null
cat $OUT/summary.txt
