#!/bin/bash
# round 5, GPU call 1: the 1024-ray reference-fixture tests; eval A/B of the kernarg layout (mode-4 removal); training A/B of the
# tiled sigma' array.  usage (gpurun): bash profiles/r05_run1.sh
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r05; mkdir -p $OUT
V=nrhints_amd/lib/variants
timeout 900 python -m pytest tests/test_gpu_train1024.py tests/test_gpu_split.py tests/test_gpu_train_fused.py tests/test_gpu_wide.py -x -q 2>&1 | tail -25 > $OUT/run1_tests.log
for i in 1 2; do
  for v in default argpad; do
    lib=""; [ $v != default ] && lib=$PWD/$V/libnrh_$v.so
    echo "== $v $i" >> $OUT/eval_ab.log
    NRHINTS_HIP_LIB=$lib timeout 300 python bench.py --steps 4 --warmup 2 --cpu-rays 0 --no-train --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])" >> $OUT/eval_ab.log
  done
done
for i in 1 2 3; do
  for v in default s1rows; do
    lib=""; [ $v != default ] && lib=$PWD/$V/libnrh_$v.so
    echo "== $v $i" >> $OUT/train_s1_ab.log
    NRHINTS_HIP_LIB=$lib timeout 300 python profiles/train_bench.py 1024 40 graph 2>/dev/null | tail -1 >> $OUT/train_s1_ab.log
  done
done
cat $OUT/run1_tests.log $OUT/eval_ab.log $OUT/train_s1_ab.log
