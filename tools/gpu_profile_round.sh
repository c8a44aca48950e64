#!/usr/bin/env bash
# The round's judged profiles of bench.py's default configuration on the GPU box: tools/gpu_profile_round.sh <tag>   (outputs under gpurun_out/<tag>/; copy what
# is to be judged into profiles/).  Kernel statistics and counters are separate runs (gpurun refuses --pmc next to the hip / hsa trace domains).
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r03_prof}; O=gpurun_out/$T; mkdir -p $O
CMD="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --no-converging --sustain-s 0"
rm -rf /tmp/prof_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- $CMD > $O/bench_under_profiler.json 2> $O/bench_under_profiler.err
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats.csv
python tools/kernel_stats.py /tmp/prof_stats "rocprofv3 --kernel-trace --stats -- $CMD" > $O/kernel_stats.txt
# SQ counters of the bench command (the TCC passes — FETCH_SIZE / WRITE_SIZE — of this command do not finish; tools/gpu_pmc_traffic.sh takes them on the batch alone)
rm -rf /tmp/prof_pmc_3
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d /tmp/prof_pmc_3 -- $CMD > /dev/null 2> $O/pmc_3.err
python tools/pmc_summary.py /tmp/prof_pmc_3 > $O/pmc_sq.txt
bash tools/gpu_pmc_traffic.sh $T 4096 > /dev/null 2>&1
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --total-pairs 1024 --no-cpu-baseline > $O/bench_tp1024.json 2> $O/bench_tp1024.err
timeout 600 python bench.py --total-pairs 128 --no-cpu-baseline > $O/bench_tp128.json 2> $O/bench_tp128.err
timeout 600 python bench.py --data demo --no-cpu-baseline > $O/bench_demo.json 2> $O/bench_demo.err
timeout 300 python tools/gpu_odometry.py 8 > $O/odometry_resident.txt 2>&1
timeout 300 python tools/gpu_odometry.py 8 --host > $O/odometry_host.txt 2>&1
timeout 300 python tools/gpu_icp_phases.py 256 > $O/icp_phases_256.txt 2>&1
timeout 300 python tools/gpu_cert_phases.py 4096 > $O/cert_phases_4096.txt 2>&1
head -12 $O/kernel_stats.txt; grep -E 'k_cert|k_nn_lds' $O/pmc_traffic.txt; tail -2 $O/odometry_resident.txt
