#!/usr/bin/env bash
# pytest -m gpu + the large configurations: tools/gpu_round_check.sh <tag> [pytest args]
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-check}; shift || true
O=gpurun_out/$T; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider "$@" > $O/pytest.txt 2>&1
tail -40 $O/pytest.txt
bash tools/gpu_large_round.sh $T > /dev/null 2>&1
cat $O/large.txt
