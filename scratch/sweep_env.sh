#!/bin/bash
# Runs ON the GPU box: bench.py under one environment switch at a time (same box, back to back).  Usage:
#   bash scratch/sweep_env.sh TAG "VAR=VAL [VAR2=VAL2]" "..." ...   ("" = defaults)
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p $OUT
i=0
for cfg in "$@"; do
  i=$((i+1))
  ( export $cfg; python3 $ROOT/bench.py --steps 100 --no-cpu-baseline > $OUT/run_$i.log 2>&1 )
  echo "$cfg => $(tail -1 $OUT/run_$i.log | python3 -c 'import json,sys
try:
    d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])
except Exception as e: print("FAILED")')"
done
