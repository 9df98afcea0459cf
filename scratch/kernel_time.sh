#!/bin/bash
# ON the GPU box: average duration of the kernels whose name matches $1 over `bench.py --steps 30` (kernel trace) -> stdout
set -e
PAT=$1; TAG=${2:-ktime}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 $ROOT/bench.py --steps 30 --warmup 2 --no-cpu-baseline --timer-reps 1 > $OUT/trace.log 2>&1
rm -f $OUT/trace/*kernel_trace.csv
grep -E "$PAT" $OUT/trace/t_kernel_stats.csv | awk -F'","' '{printf "%-80s calls %s avg %.1f us\n", substr($1,2,80), $2, $4/1000}' 
grep '"metric"' $OUT/trace.log | cut -c60-160
