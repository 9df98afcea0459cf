#!/bin/bash
# Runs ON the GPU box (via gpurun): kernel-trace stats of the default bench (hipGraph replay) and separate PMC
# passes over the eager run (--pmc never beside a tracing domain; the program sits directly after `--`):
#   FETCH_SIZE | WRITE_SIZE                                   -> HBM-side traffic per kernel
#   SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA GRBM_GUI_ACTIVE -> MFMA utilisation
# Output under gpurun_out/$1/.
set -e
TAG=${1:-prof2}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 $ROOT/bench.py --steps 100 --warmup 2 --no-cpu-baseline --timer-reps 1 > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log | cut -c1-200
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; cnt=${pass#*:}
  rocprofv3 --pmc $cnt -d $OUT/pmc_$name -o c --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > $OUT/pmc_$name.log 2>&1
  python3 $ROOT/scratch/pmc_summary.py $OUT/pmc_$name 0 > /dev/null
  rm -f $OUT/pmc_$name/*counter_collection.csv $OUT/pmc_$name/*/*counter_collection.csv
  echo "$name done"
done
rm -f $OUT/trace/*kernel_trace.csv $OUT/trace/*/*kernel_trace.csv
ls $OUT $OUT/pmc_mfma | head -30
