#!/usr/bin/env bash
# Same-box A/B of library builds and option presets through bench.py: tools/gpu_ab_libs.sh <tag> <rounds> "<name>=<lib or default>[,ENV=VALUE...]" ...
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=$1; R=$2; shift 2; O=gpurun_out/$T; mkdir -p $O
for r in $(seq 1 "$R"); do
	for V in "$@"; do
		N=${V%%=*}; REST=${V#*=}; L=${REST%%,*}; ENVS=""; [ "$REST" != "$L" ] && ENVS=$(echo "${REST#*,}" | tr ',' ' ')
		P=$PWD/$L; [ "$L" = default ] && P=$PWD/mulls_amd/libmulls_hip.so
		env $ENVS MULLS_HIP_LIB=$P timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-end-to-end --steps 10 --sustain-s 2 2>$O/err_$N.txt | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernel_ms_per_step']
print('%-22s value %.1f k  sustained %.1f k  converging %.1f k | search %.2f accum %.2f setup %.2f ms/step, search launch %.1f us, frac %.3f' % ('$N', j['value']/1e3, j['value_sustained']['value']/1e3, j.get('value_converging',{}).get('value',0)/1e3, k['ms_nn'], k['ms_accum'], k['ms_setup'], j['roofline']['avg_launch_ms']*1e3, j['roofline']['frac']))" | tee -a $O/ab_libs.txt
	done
done
