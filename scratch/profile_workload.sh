#!/bin/bash
# ON the GPU box: kernel-trace stats of `bench.py --workload $1` (hipGraph replay, 50 timed steps) -> gpurun_out/$2/
set -e
WL=$1; TAG=$2
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 $ROOT/bench.py --workload $WL --steps 50 --warmup 2 --no-cpu-baseline > $OUT/trace.log 2>&1
grep '"metric"' $OUT/trace.log | cut -c1-220
rm -f $OUT/trace/*kernel_trace.csv $OUT/trace/*/*kernel_trace.csv
