#!/bin/bash
# usage: pmc_harness.sh <tag> <binary> [args...]   (on the GPU box, from the repo root) -> gpurun_out/pmc_<tag>/summary.txt
# rocprofv3 --pmc passes over a stand-alone binary (profiles/ubench/data/sdf32_bench ...): SQ issue/wait split, instruction mix,
# LDS conflicts, clock.  One pass per counter group (8 SQ slots).
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
BIN=$GRAFT_REPO_ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- $BIN $ARGS > $OUT/$n.log 2>&1; }
ARGS="$*"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD
run tcc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 - "$OUT" > $OUT/summary.txt <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        dur[row["Kernel_Name"][:60]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
for k in sorted(agg):
    d = dur.get(k, [0])
    print(k, f"  launches/pass {len(d)//5 if d else 0}  mean ms (profiled) {sum(d)/max(1,len(d)):.3f}")
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
PY
cat $OUT/summary.txt
