#!/usr/bin/env bash
# k-candidate certificates on / off (and the list length from which the LDS tier takes the look) on the same box: tools/gpu_kcert_ab.sh <tag> [rounds] ["0:192 1:192 1:64 ..."]
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-kcert}; R=${2:-2}; V=${3:-"0:192 1:192"}; O=gpurun_out/$T; mkdir -p $O
for r in $(seq 1 "$R"); do
	for KV in $V; do
		K=${KV%%:*}; M=${KV##*:}
		MULLS_KCERT=$K MULLS_KCERT_MIN=$M timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-end-to-end --steps 10 --sustain-s 2 2>$O/err_$K.txt | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernel_ms_per_step']
print('kcert=$K min=$M  value %.1f k  sustained %.1f k  converging %.1f k | search %.2f accum %.2f setup %.2f ms/step, search launch %.1f us, frac %.3f' % (j['value']/1e3, j['value_sustained']['value']/1e3, j.get('value_converging',{}).get('value',0)/1e3, k['ms_nn'], k['ms_accum'], k['ms_setup'], j['roofline']['avg_launch_ms']*1e3, j['roofline']['frac']))" | tee -a $O/ab.txt
	done
done
if [ "${4:-}" = phases ]; then
for KV in $V; do K=${KV%%:*}; M=${KV##*:}; echo "== MULLS_KCERT=$K MULLS_KCERT_MIN=$M" | tee -a $O/ab.txt; MULLS_KCERT=$K MULLS_KCERT_MIN=$M timeout 300 python tools/gpu_cert_phases.py 4096 2>&1 | tee -a $O/ab.txt; done
fi
