#!/bin/bash
# Runs ON the GPU box: same-box A/B of two builds of libvmr_hip.so (vmrframe_amd/lib/libvmr_hip_old.so = the build to
# compare against, made by hand from an older checkout of csrc/) -- bench line and rocprofv3 kernel stats of each.
set -e
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/${1:-ab}
L=$ROOT/vmrframe_amd/lib
mkdir -p $OUT
cp $L/libvmr_hip.so /tmp/new.so
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for v in new old; do
    if [ $v = new ]; then cp /tmp/new.so $L/libvmr_hip.so; else cp $L/libvmr_hip_old.so $L/libvmr_hip.so; fi
    python3 $ROOT/bench.py --steps 100 --no-cpu-baseline > $OUT/bench_${v}_$rep.log 2>&1
    tail -1 $OUT/bench_${v}_$rep.log | cut -c1-120
  done
done
for v in new old; do
  if [ $v = new ]; then cp /tmp/new.so $L/libvmr_hip.so; else cp $L/libvmr_hip_old.so $L/libvmr_hip.so; fi
  rocprofv3 --kernel-trace --stats -d $OUT/trace_$v -o t --output-format csv -- python3 $ROOT/bench.py --steps 100 --warmup 2 --no-cpu-baseline > $OUT/trace_$v.log 2>&1
  rm -f $OUT/trace_$v/*kernel_trace.csv $OUT/trace_$v/*/*kernel_trace.csv
done
cp /tmp/new.so $L/libvmr_hip.so
echo done
