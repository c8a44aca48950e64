cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_final3; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
CMD="python bench.py --no-other-configs --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --no-converging --sustain-s 0"
rm -rf /tmp/prof_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- $CMD > $O/bench_under_profiler.json 2> $O/bench_under_profiler.err
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats.csv
python tools/kernel_stats.py /tmp/prof_stats "rocprofv3 --kernel-trace --stats -- $CMD" > $O/kernel_stats.txt
timeout 500 python bench.py > $O/bench.json 2> $O/bench.err
head -6 $O/kernel_stats.txt
python -c "
import json; j=json.load(open('$O/bench.json')); print('bench', round(j['value'],1), 'ms/step', round(j['ms_per_step'],3), 'frac', round(j['roofline']['frac'],4), 'sust', round(j['value_sustained']['value']), 'conv', round(j['value_converging']['value']))"
