#!/usr/bin/env bash
# Kernel timeline of small batches on the lock-step path: tools/gpu_small_trace.sh <tag> [pairs ...]   (kernel durations + gaps between consecutive kernels)
set -u
TAG=${1:-small}; shift || true
SIZES=${*:-"1 128"}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for n in $SIZES; do
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$n -- python tools/gpu_icp_phases_lock.py $n 6 > $OUT/trace_$n.log 2>&1
  python tools/kernel_stats.py $OUT/trace_$n > $OUT/stats_$n.txt 2>&1
  python tools/kernel_gaps.py $OUT/trace_$n > $OUT/gaps_$n.txt 2>&1
done
tail -n 40 $OUT/stats_*.txt $OUT/gaps_*.txt
