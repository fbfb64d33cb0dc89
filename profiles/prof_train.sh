#!/bin/bash
# rocprofv3 kernel statistics of the training step (profiles/train_bench.py) -> gpurun_out/prof_train_<tag>/
# usage: bash profiles/prof_train.sh <tag> <batch> <sdf_backward>
set -e
TAG=${1:-hip}; BATCH=${2:-1024}; IMPL=${3:-hip}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_train_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o train -- python $REPO/profiles/train_bench.py $BATCH 10 $IMPL > $OUT/run.log 2>&1
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
python - "$f" > $OUT/summary.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over the run")
for r in rows[:40]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.3f} ms {float(r["TotalDurationNs"])/tot*100:5.1f}%  calls {r["Calls"]:>6}  avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:110]}')
PY
