#!/usr/bin/env bash
# Same-box A/B of library builds on one bench configuration: tools/gpu_ab_cfg.sh <tag> <rounds> "<bench.py arguments>" "<name>=<lib or default>[,ENV=VALUE...]" ...
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=$1; R=$2; ARGS=$3; shift 3; O=gpurun_out/$T; mkdir -p $O
for r in $(seq 1 "$R"); do
	for V in "$@"; do
		N=${V%%=*}; REST=${V#*=}; L=${REST%%,*}; ENVS=""; [ "$REST" != "$L" ] && ENVS=$(echo "${REST#*,}" | tr ',' ' ')
		P=$PWD/$L; [ "$L" = default ] && P=$PWD/mulls_amd/libmulls_hip.so
		env $ENVS MULLS_HIP_LIB=$P timeout 300 python bench.py $ARGS --no-other-configs --no-cpu-baseline --no-end-to-end --no-converging --sustain-s 0 2>$O/err_$N.txt | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); rf=j.get('roofline',{}); k=rf.get('kernel_ms_per_step',{})
print('%-14s value %10.1f  ms/step %8.3f | search %.2f accum %.2f setup %.2f ms/step, search launch %.1f us, frac %.3f' % ('$N', j['value'], j['ms_per_step'], k.get('ms_nn',0), k.get('ms_accum',0), k.get('ms_setup',0), rf.get('avg_launch_ms',0)*1e3, rf.get('frac',0)))" | tee -a $O/ab.txt
	done
done
