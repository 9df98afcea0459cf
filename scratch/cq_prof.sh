#!/bin/bash
# ON the GPU box: per-kernel durations of the stand-alone CQ block (scratch/cq_bench.py) -> gpurun_out/$1/
set -e
TAG=$1
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 $ROOT/scratch/cq_bench.py 50 > $OUT/trace.log 2>&1
rm -f $OUT/trace/*kernel_trace.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/trace/t_kernel_stats.csv")))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print(f"{r['Name'][:90]:90s} x{r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:7.1f} us")
PY
