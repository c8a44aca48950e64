#!/usr/bin/env bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_g; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
timeout 300 python tools/gpu_odometry.py 8 --motion-compensation 1 --check 3 > $O/odometry_mc.txt 2>&1; tail -3 $O/odometry_mc.txt
bash tools/gpu_large_round.sh r04_g > /dev/null 2>&1
grep "per run" $O/large.txt
for n in 16; do timeout 300 python tools/gpu_large_bench.py cfg4 $n 3 2>&1 | grep -v Warn | head -3; done
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; j=json.load(open('$O/bench.json')); print('bench', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernel_ms_per_step'], j.get('value_converging',{}).get('value'), j.get('value_end_to_end',{}).get('calls_ms'))"
