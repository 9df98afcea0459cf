#!/bin/bash
# Runs ON the GPU box (via gpurun): kernel-trace stats of the default bench (hipGraph replay) and two
# separate PMC passes (FETCH_SIZE, WRITE_SIZE) of an eager run.  Output under gpurun_out/$1/.
set -e
TAG=${1:-prof}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log | cut -c1-160
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o w --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > $OUT/pmc_write.log 2>&1
echo write done
# keep only compact summaries of the counter files (they are large)
python3 $ROOT/scratch/summarize_profile.py $OUT
rm -f $OUT/pmc_fetch/*counter_collection.csv $OUT/pmc_write/*counter_collection.csv $OUT/trace/*kernel_trace.csv
ls $OUT
