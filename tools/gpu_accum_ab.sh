#!/usr/bin/env bash
# normal equations by one wave per trip (k_accum_wave) against one workgroup per trip (k_accum), same box: tools/gpu_accum_ab.sh <tag> [rounds]
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-accum}; R=${2:-2}; O=gpurun_out/$T; mkdir -p $O
for r in $(seq 1 "$R"); do
	for W in 0 2048; do
		MULLS_ACCUM_WAVE_MIN_TRIPS=$W timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-end-to-end --steps 10 --sustain-s 2 2>$O/err_$W.txt | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernel_ms_per_step']
print('wave_min_trips=$W  value %.1f k  sustained %.1f k  converging %.1f k | search %.2f accum %.2f setup %.2f ms/step, search launch %.1f us, frac %.3f  table %s' % (j['value']/1e3, j['value_sustained']['value']/1e3, j.get('value_converging',{}).get('value',0)/1e3, k['ms_nn'], k['ms_accum'], k['ms_setup'], j['roofline']['avg_launch_ms']*1e3, j['roofline']['frac'], str(j.get('result_table_sha256'))[:16]))" | tee -a $O/ab_accum.txt
	done
done
