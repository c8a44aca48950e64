#!/usr/bin/env bash
# The full-length bench lines of the other BASELINE configurations: tools/gpu_cfg_lines.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-cfg}; O=gpurun_out/$T; mkdir -p $O
timeout 300 python bench.py --config 0 --no-cpu-baseline > $O/bench_cfg0.json 2> $O/bench_cfg0.err
timeout 300 python bench.py --config 3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err
timeout 300 python bench.py --config 2 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 300 python bench.py --config 4 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
for f in bench_cfg0 bench_cfg3 bench_cfg2 bench_cfg4; do python -c "
import json; j=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', round(j['value'],1), j['unit'], 'ms/step', round(j['ms_per_step'],3), 'frac', round(j['roofline']['frac'],4), 'conv', (j.get('value_converging') or {}).get('value'))"; done
