#!/bin/bash
# ON the GPU box: one rocprofv3 counter pass (no tracing domains beside it).  usage: pmc_pass.sh TAG "COUNTERS" script [args]
TAG=$1; CNT=$2; shift; shift
ROOT=$GRAFT_REPO_ROOT
mkdir -p $ROOT/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT -d $ROOT/gpurun_out/$TAG -o c --output-format csv -- python3 $ROOT/"$@" > $ROOT/gpurun_out/$TAG/run.log 2>&1
python3 $ROOT/scratch/pmc_summary.py $ROOT/gpurun_out/$TAG
rm -f $ROOT/gpurun_out/$TAG/*counter_collection.csv $ROOT/gpurun_out/$TAG/*/*counter_collection.csv
