set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06k
( timeout 900 python -m pytest tests/test_gpu_wide.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r06k/tests_wide.log 2>&1
for i in 1 2; do
 for v in mix nomix; do
  if [ $v = nomix ]; then export NRHINTS_HIP_LIB=$PWD/nrhints_amd/lib/variants/libnrh_nomix.so; else unset NRHINTS_HIP_LIB; fi
  timeout 600 python bench.py --steps 4 --warmup 1 --cpu-rays 0 --no-train --no-secondary --no-camopt 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        j = json.loads(l); print('$v', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['avg_launch_ms'], j.get('reduced', {}).get('value'))
" >> gpurun_out/r06k/ab_mix.log
 done
done
cat gpurun_out/r06k/tests_wide.log gpurun_out/r06k/ab_mix.log
