#!/usr/bin/env bash
# HBM traffic counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) of the bench batch through the lock-step path:
#   tools/gpu_pmc_traffic.sh <tag> [pairs]     -> gpurun_out/<tag>/pmc_traffic.txt
# (the batch alone, tools/gpu_icp_phases_lock.py: the counter passes of the whole bench.py command did not finish within 15 minutes each)
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r03_pmc}; N=${2:-4096}; O=gpurun_out/$T; mkdir -p $O
i=0
for c in FETCH_SIZE WRITE_SIZE; do
	i=$((i + 1)); rm -rf /tmp/pmc_t_$i
	MULLS_SPLIT_MAX_PAIRS=0 timeout ${PMC_TIMEOUT:-150} rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_t_$i -- python tools/gpu_icp_phases_lock.py $N 2 > $O/pmc_run_$i.log 2>&1
done
python tools/pmc_summary.py /tmp/pmc_t_1 /tmp/pmc_t_2 > $O/pmc_traffic.txt
cat $O/pmc_traffic.txt
