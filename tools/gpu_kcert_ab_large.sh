#!/usr/bin/env bash
# k-candidate certificates on / off on the large configurations and the scan-to-map batch, same box: tools/gpu_kcert_ab_large.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-kcert}; O=gpurun_out/$T; mkdir -p $O
run() { timeout 300 python tools/gpu_large_bench.py "$@" 2>&1 | grep -v Warning | grep -E "pairs:|ms per run|kernel ms"; }
for c in "cfg2 32 3" "cfg4 16 3" "s2m 64 5" "s2mcap 64 5" "cfg2 1 5" "cfg4 1 5"; do
	for K in 0 1; do echo "== MULLS_KCERT=$K $c"; MULLS_KCERT=$K run $c; done
done 2>&1 | tee $O/ab_large.txt
