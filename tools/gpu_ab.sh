#!/usr/bin/env bash
# A/B timing of two builds of the library on the same box: tools/gpu_ab.sh <lib_a.so> <lib_b.so> [rounds]
# (k_nn_lds average per launch, 1024 pairs, one launch per iteration; alternating runs so that drift hits both alike)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${3:-2}
for r in $(seq 1 "$R"); do
	for v in a b; do
		L=$1; [ $v = b ] && L=$2
		MULLS_HIP_LIB=$PWD/$L MULLS_SUBBATCHES=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ab_${v}_$r -- python tools/gpu_one.py 0 1024 2 >/dev/null 2>&1
		echo "$v $r $(python tools/kernel_stats.py gpurun_out/ab_${v}_$r | sed -n 2,4p | tr "\n" "|")"
	done
done
