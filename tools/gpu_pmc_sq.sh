#!/usr/bin/env bash
# SQ counters (one rocprofv3 --pmc pass, no trace domains) of the bench batch through the lock-step path: tools/gpu_pmc_sq.sh <tag> [pairs] [ENV=VALUE ...]  -> gpurun_out/<tag>/pmc_sq.txt
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-pmc}; N=${2:-4096}; shift 2 || true
O=gpurun_out/$T; mkdir -p $O
rm -rf /tmp/pmc_sq
env "$@" MULLS_SPLIT_MAX_PAIRS=0 timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d /tmp/pmc_sq -- python tools/gpu_icp_phases_lock.py $N 2 > $O/pmc_sq_run.log 2>&1
{ echo "# rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAVES -- python tools/gpu_icp_phases_lock.py $N 2   ($*)"; echo "# averages per launch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles summed over waves (MI355X_MICROARCH.md)"; python tools/pmc_summary.py /tmp/pmc_sq; } > $O/pmc_sq.txt
cat $O/pmc_sq.txt
