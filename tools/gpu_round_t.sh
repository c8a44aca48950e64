cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r03_t; mkdir -p $O
rm -rf /tmp/prof_b
MULLS_BENCH_TRACE=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_b -- python bench.py --no-converging --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
grep "^step" $O/bench.err
python tools/big_gaps.py /tmp/prof_b 5000 > $O/gaps.txt; tail -40 $O/gaps.txt
