#!/bin/bash
# ON the GPU box: kernel-trace stats of the default bench (hipGraph replay).  usage: prof_trace.sh TAG [bench args]
TAG=$1; shift
ROOT=$GRAFT_REPO_ROOT
mkdir -p $ROOT/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/$TAG -o t --output-format csv -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $ROOT/gpurun_out/$TAG/trace.log 2>&1
rm -f $ROOT/gpurun_out/$TAG/*kernel_trace.csv $ROOT/gpurun_out/$TAG/*/*kernel_trace.csv
grep -o '"value": [0-9.]*, "unit": "clips/sec", "n_gpus": 1, "steps": 10, "warmup": 2, "ms_per_step": [0-9.]*' $ROOT/gpurun_out/$TAG/trace.log
