#!/usr/bin/env bash
# SQ counters of the bench batch through the lock-step path, TWO counters per rocprofv3 --pmc pass (a pass of eight did not return on this pool within 200 s; passes of
# one or two take ~25 s): tools/gpu_pmc_sq.sh <tag> [pairs] [ENV=VALUE ...]  -> gpurun_out/<tag>/pmc_sq.txt
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-pmc}; N=${2:-4096}; shift 2 || true
O=gpurun_out/$T; mkdir -p $O
dirs=""
i=0
for pair in "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_SALU SQ_BUSY_CYCLES"; do
	i=$((i + 1)); rm -rf /tmp/pmc_sq_$i
	env "$@" MULLS_SPLIT_MAX_PAIRS=0 timeout 100 rocprofv3 --pmc $pair --output-format csv -d /tmp/pmc_sq_$i -- python tools/gpu_icp_phases_lock.py $N 2 > $O/pmc_sq_run_$i.log 2>&1 || { echo "pass $i ($pair) did not return" >> $O/pmc_sq_failed.txt; break; }
	dirs="$dirs /tmp/pmc_sq_$i"
done
{ echo "# rocprofv3 --pmc <two SQ counters per pass> -- python tools/gpu_icp_phases_lock.py $N 2   ($*)"; echo "# averages per launch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles summed over waves (MI355X_MICROARCH.md)"; python tools/pmc_summary.py $dirs; } > $O/pmc_sq.txt
cat $O/pmc_sq.txt
