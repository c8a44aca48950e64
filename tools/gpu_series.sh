#!/usr/bin/env bash
# Launch-by-launch durations of the search kernels on the bench workload: tools/gpu_series.sh <tag> [pairs]
set -u
TAG=${1:-series}; NB=${2:-4096}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- env MULLS_SPLIT_MAX_PAIRS=0 python tools/gpu_icp_phases_lock.py $NB 3 > $OUT/run.log 2>&1
for k in "void k_cert" k_nn_lds "void k_accum<1024" k_tgt_grid; do echo "-- $k"; python tools/nn_series.py $OUT/trace "$k" 2>&1 | tail -4; done > $OUT/series.txt
cat $OUT/series.txt
