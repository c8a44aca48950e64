#!/usr/bin/env bash
# Same-box A/B of library builds on configs[0] (the reference's demo scans): tools/gpu_ab_cfg0.sh <rounds> <lib> ...
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$1; shift
for r in $(seq 1 "$R"); do for L in "$@"; do
MULLS_HIP_LIB=$PWD/$L timeout 300 python bench.py --config 0 --no-cpu-baseline --no-end-to-end --steps 10 --sustain-s 1 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernel_ms_per_step']
print('%-36s value %.1f k ms/step %.2f search %.2f accum %.2f setup %.2f' % ('$L', j['value']/1e3, j['ms_per_step'], k['ms_nn'], k['ms_accum'], k['ms_setup']))"
done; done
