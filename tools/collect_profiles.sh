#!/usr/bin/env bash
# Copy what is judged from a gpu_profile_round.sh run into profiles/: tools/collect_profiles.sh <gpurun_out tag> <round prefix, e.g. r06> [tag of a separate WRITE_SIZE pass]
set -eu
cd "$(dirname "$0")/.."
S=gpurun_out/$1; P=profiles/$2; W=${3:-}
cp $S/kernel_stats.txt ${P}_kernel_stats.txt
cp $S/rocprofv3_kernel_stats.csv ${P}_rocprofv3_kernel_stats.csv
for c in 2 4; do cp $S/kernel_stats_bench_cfg$c.txt ${P}_kernel_stats_bench_cfg$c.txt; done
{ echo "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python tools/gpu_icp_phases_lock.py 4096 2   (KB per launch, averages over the launches of two runs)"; cat $S/pmc_traffic.txt; if [ -n "$W" ]; then echo "# WRITE_SIZE pass taken again on its own (the pass inside the round's call did not return: gpurun_out/$W)"; cat gpurun_out/$W/pmc_write.txt; fi; } > ${P}_pmc_traffic.txt
cp $S/pmc_traffic_cfg4.txt ${P}_pmc_traffic_cfg4.txt
[ -f $S/pmc_sq.txt ] && cp $S/pmc_sq.txt ${P}_pmc_sq.txt
cp $S/bench.json ${P}_bench.json
cp $S/bench_under_profiler.json ${P}_bench_under_profiler.json
for c in cfg0 cfg2 cfg3 cfg4 tp128; do cp $S/bench_$c.json ${P}_bench_$c.json; done
cp $S/modes.txt ${P}_modes.txt
{ cat $S/large.txt; for c in cfg4 cfg2 s2m; do echo; cat $S/kernel_stats_$c.txt; done; } > ${P}_large.txt
cp $S/odometry_resident.txt ${P}_odometry.txt
cp $S/odometry_motion_compensation.txt ${P}_odometry_motion_compensation.txt
cp $S/e2e_calls.txt ${P}_e2e_calls.txt
ls ${P}_*
