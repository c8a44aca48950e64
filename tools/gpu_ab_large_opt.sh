#!/usr/bin/env bash
# An option on / off on the large configurations, same box: tools/gpu_ab_large_opt.sh <tag> <ENVNAME> "<case pairs reps>" ...
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=$1; E=$2; shift 2; O=gpurun_out/$T; mkdir -p $O
run() { timeout 300 python tools/gpu_large_bench.py "$@" 2>&1 | grep -v Warning | grep -E "ms per run|kernel ms|integer outputs"; }
for c in "$@"; do
	for K in 0 1; do echo "== $E=$K $c"; env $E=$K bash -c "$(declare -f run); run $c"; done
done 2>&1 | tee $O/ab_large_$E.txt
