#!/usr/bin/env bash
# The large configurations on the GPU box: tools/gpu_large_round.sh <tag> [quick]   (outputs under gpurun_out/<tag>/)
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-large}; O=gpurun_out/$T; mkdir -p $O
run() { timeout 300 python tools/gpu_large_bench.py "$@" 2>&1 | grep -v Warning; }
{
run cfg4 1 5 --check
run cfg4 8 3
run cfg2 1 5 --check
run cfg2 8 3
run s2mcap 64 5
run s2m 64 5 --check
run s2m 1 5
run s2mcap 1 5
} > $O/large.txt 2>&1
for c in cfg4 cfg2; do
  rm -rf /tmp/prof_$c
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$c -- python tools/gpu_large_bench.py $c 1 5 > $O/prof_$c.out 2>&1
  python tools/kernel_stats.py /tmp/prof_$c "rocprofv3 --kernel-trace --stats -- python tools/gpu_large_bench.py $c 1 5" > $O/kernel_stats_$c.txt 2>&1
  python tools/nn_series.py /tmp/prof_$c k_cert_big 2>&1 | tail -3 >> $O/kernel_stats_$c.txt
done
rm -rf /tmp/prof_s2m
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s2m -- python tools/gpu_large_bench.py s2m 64 5 > $O/prof_s2m.out 2>&1
python tools/kernel_stats.py /tmp/prof_s2m "rocprofv3 --kernel-trace --stats -- python tools/gpu_large_bench.py s2m 64 5" > $O/kernel_stats_s2m.txt 2>&1
python tools/nn_series.py /tmp/prof_s2m k_cert_big 2>&1 | tail -3 >> $O/kernel_stats_s2m.txt
python tools/nn_series.py /tmp/prof_s2m k_cert_nn 2>&1 | tail -3 >> $O/kernel_stats_s2m.txt
cat $O/large.txt; head -30 $O/kernel_stats_cfg4.txt
