#!/usr/bin/env bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_f; mkdir -p $O
timeout 300 python tools/gpu_odometry.py 8 --motion-compensation 1 --check 3 > $O/odometry_mc.txt 2>&1; tail -3 $O/odometry_mc.txt
rm -rf /tmp/prof_a; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -- python tools/gpu_large_bench.py cfg4 16 3 > $O/prof_cfg4x16.out 2>&1
python tools/kernel_stats.py /tmp/prof_a "cfg4 x 16" > $O/kernel_stats_cfg4x16.txt 2>&1
for k in k_cert_big k_filter "void k_accum" k_finish k_step; do echo "-- $k"; python tools/nn_series.py /tmp/prof_a "$k" | tail -2; done >> $O/kernel_stats_cfg4x16.txt 2>&1
cat $O/kernel_stats_cfg4x16.txt
rm -rf /tmp/prof_b; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -- python tools/gpu_large_bench.py cfg2 32 3 > $O/prof_cfg2x32.out 2>&1
python tools/kernel_stats.py /tmp/prof_b "cfg2 x 32" > $O/kernel_stats_cfg2x32.txt 2>&1
for k in k_cert_big k_filter; do echo "-- $k"; python tools/nn_series.py /tmp/prof_b "$k" | tail -2; done >> $O/kernel_stats_cfg2x32.txt 2>&1
cat $O/kernel_stats_cfg2x32.txt
