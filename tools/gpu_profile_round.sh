#!/usr/bin/env bash
# SKIP_PMC=1: without the counter passes (they hang now and then on this pool: 150 s each)
# The round's judged profiles on the GPU box: tools/gpu_profile_round.sh <tag>   (outputs under gpurun_out/<tag>/; copy what is to be judged into profiles/).
# Kernel statistics and counters are separate runs (gpurun refuses --pmc next to the hip / hsa trace domains).
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r04_prof}; O=gpurun_out/$T; mkdir -p $O
# --- configs[1], the headline line: kernel statistics of the bench command, SQ counters, FETCH / WRITE passes of the batch alone
CMD="python bench.py --no-other-configs --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --no-converging --sustain-s 0"
rm -rf /tmp/prof_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- $CMD > $O/bench_under_profiler.json 2> $O/bench_under_profiler.err
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats.csv
python tools/kernel_stats.py /tmp/prof_stats "rocprofv3 --kernel-trace --stats -- $CMD" > $O/kernel_stats.txt
# (SQ counters: tools/gpu_pmc_sq.sh, two counters per pass — passes of eight do not return on this pool)
[ "${SKIP_PMC:-0}" = 1 ] || bash tools/gpu_pmc_traffic.sh $T 4096 > /dev/null 2>&1
# --- configs[4] and configs[2]: kernel statistics of the bench command, FETCH / WRITE passes of the batch alone
for c in 4 2; do
	CMDC="python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-converging"
	rm -rf /tmp/prof_c$c
	timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c$c -- $CMDC > $O/bench_cfg${c}_under_profiler.json 2> $O/bench_cfg${c}_under_profiler.err
	python tools/kernel_stats.py /tmp/prof_c$c "rocprofv3 --kernel-trace --stats -- $CMDC" > $O/kernel_stats_bench_cfg$c.txt
done
if [ "${SKIP_PMC:-0}" != 1 ]; then
i=0
for cnt in FETCH_SIZE WRITE_SIZE; do
	i=$((i + 1)); rm -rf /tmp/pmc_c4_$i
	timeout 200 rocprofv3 --pmc $cnt --output-format csv -d /tmp/pmc_c4_$i -- python tools/gpu_large_bench.py cfg4 16 1 > $O/pmc_cfg4_run_$i.log 2>&1
done
python tools/pmc_summary.py /tmp/pmc_c4_1 /tmp/pmc_c4_2 > $O/pmc_traffic_cfg4.txt
fi
# --- the bench lines of every configuration
timeout 500 python bench.py > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --config 0 --no-cpu-baseline > $O/bench_cfg0.json 2> $O/bench_cfg0.err
timeout 600 python bench.py --config 3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err
timeout 600 python bench.py --total-pairs 128 --no-cpu-baseline > $O/bench_tp128.json 2> $O/bench_tp128.err
timeout 600 python bench.py --config 2 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 600 python bench.py --config 4 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
# --- batch sizes, large configurations, odometry
timeout 600 python tools/gpu_modes.py 1 8 32 64 128 256 512 1024 2048 4096 > $O/modes.txt 2>&1
bash tools/gpu_large_round.sh $T > /dev/null 2>&1
timeout 300 python tools/gpu_odometry.py 12 > $O/odometry_resident.txt 2>&1
timeout 300 python tools/gpu_odometry.py 12 --motion-compensation 1 --check 3 > $O/odometry_motion_compensation.txt 2>&1
timeout 300 python tools/gpu_e2e_calls.py 24 > $O/e2e_calls.txt 2>&1
head -12 $O/kernel_stats.txt; grep -E 'k_cert|k_nn_lds' $O/pmc_traffic.txt; grep -E 'k_cert|k_filter' $O/pmc_traffic_cfg4.txt; cat $O/modes.txt; tail -1 $O/odometry_resident.txt
for f in bench bench_cfg0 bench_cfg3 bench_tp128 bench_cfg2 bench_cfg4; do python -c "
import json; j=json.load(open('$O/$f.json')); print('$f', round(j['value'],1), j['unit'], 'ms/step', round(j['ms_per_step'],3), 'frac', round(j['roofline']['frac'],4))"; done
