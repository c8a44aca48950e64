#!/usr/bin/env bash
# Library builds on the large configurations, same box, alternating: tools/gpu_ab_large_libs.sh <tag> <rounds> "<lib or default> ..." "<case pairs reps [--check]>" ...
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=$1; R=$2; LIBS=$3; shift 3; O=gpurun_out/$T; mkdir -p $O
for c in "$@"; do
	for r in $(seq 1 "$R"); do
		for L in $LIBS; do
			P=$PWD/$L; [ "$L" = default ] && P=$PWD/mulls_amd/libmulls_hip.so
			echo "== $(basename $L .so) | $c"
			MULLS_HIP_LIB=$P timeout 300 python tools/gpu_large_bench.py $c 2>&1 | grep -v Warning | grep -E "ms per run|kernel ms|integer outputs"
		done
	done
done 2>&1 | tee $O/ab_large.txt
