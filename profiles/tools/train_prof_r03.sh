R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03/final2; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_wide.py tests/test_gpu_parity2.py tests/test_gpu_parity.py -q -k "raygen or eval_dicts or wide or render_image or register_view or golden or fused_head or ray_generator_group" > $OUT/pytest_subset.log 2>&1; echo rc=$? >> $OUT/pytest_subset.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o train -- python $R/profiles/train_bench.py 1024 10 graph > $OUT/prof_train.log 2>&1
f=$(find $OUT/prof_train -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/rocprof_train_stats.csv
f=$(find $OUT/prof_train -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && python $R/profiles/step_breakdown.py $f detail > $OUT/train_step_breakdown.txt 2>&1
rm -rf $OUT/prof_train
