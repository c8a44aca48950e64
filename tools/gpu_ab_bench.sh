#!/usr/bin/env bash
# Same-box A/B of library builds through bench.py (4096 pairs): tools/gpu_ab_bench.sh <rounds> <lib.so> [<lib.so> ...]   ("default" = the in-tree build)
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$1; shift
for r in $(seq 1 "$R"); do
	for L in "$@"; do
		P=$PWD/$L; [ "$L" = default ] && P=$PWD/mulls_amd/libmulls_hip.so
		MULLS_HIP_LIB=$P timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-end-to-end --steps 10 --sustain-s 2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernel_ms_per_step']
print('%-28s value %.1f k  sustained %.1f k  converging %.1f k | search %.2f accum %.2f setup %.2f ms/step, search launch %.1f us' % ('$L', j['value']/1e3, j['value_sustained']['value']/1e3, j.get('value_converging',{}).get('value',0)/1e3, k['ms_nn'], k['ms_accum'], k['ms_setup'], j['roofline']['avg_launch_ms']*1e3))"
	done
done
