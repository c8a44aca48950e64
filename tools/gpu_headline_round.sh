#!/usr/bin/env bash
# The headline part of gpu_profile_round.sh alone (bench line, kernel statistics of the bench command, batch sizes, odometry): tools/gpu_headline_round.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r04_head}; O=gpurun_out/$T; mkdir -p $O
CMD="python bench.py --no-other-configs --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --no-converging --sustain-s 0"
rm -rf /tmp/prof_stats
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- $CMD > $O/bench_under_profiler.json 2> $O/bench_under_profiler.err
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats.csv
python tools/kernel_stats.py /tmp/prof_stats "rocprofv3 --kernel-trace --stats -- $CMD" > $O/kernel_stats.txt
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python tools/gpu_modes.py 1 8 32 64 128 256 512 1024 2048 4096 > $O/modes.txt 2>&1
timeout 120 python tools/gpu_odometry.py 12 > $O/odometry_resident.txt 2>&1
timeout 120 python tools/gpu_odometry.py 12 --motion-compensation 1 --check 3 > $O/odometry_motion_compensation.txt 2>&1
timeout 200 python bench.py --total-pairs 128 --no-cpu-baseline > $O/bench_tp128.json 2> $O/bench_tp128.err
head -12 $O/kernel_stats.txt; cat $O/modes.txt; tail -1 $O/odometry_resident.txt
for f in bench bench_tp128; do python -c "
import json; j=json.load(open('$O/$f.json')); print('$f', round(j['value'],1), j['unit'], 'ms/step', round(j['ms_per_step'],3), 'frac', round(j['roofline']['frac'],4), 'conv', j.get('value_converging',{}).get('value'), 'e2e', j.get('value_end_to_end',{}).get('value'))"; done
