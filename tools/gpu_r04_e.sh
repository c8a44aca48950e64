#!/usr/bin/env bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_e; mkdir -p $O
timeout 600 python -m pytest tests/test_motion_comp.py tests/test_gpu_adapter.py -m gpu -q -p no:cacheprovider > $O/pytest_mc.txt 2>&1; tail -5 $O/pytest_mc.txt
timeout 300 python tools/gpu_odometry.py 8 --motion-compensation 1 --check 3 > $O/odometry_mc.txt 2>&1; tail -3 $O/odometry_mc.txt
timeout 300 python tools/gpu_odometry.py 8 > $O/odometry.txt 2>&1; tail -2 $O/odometry.txt
timeout 600 python bench.py --config 4 > $O/bench_cfg4.json 2> $O/bench_cfg4.err; tail -c 600 $O/bench_cfg4.json; tail -3 $O/bench_cfg4.err
timeout 600 python bench.py --config 2 > $O/bench_cfg2.json 2> $O/bench_cfg2.err; tail -c 600 $O/bench_cfg2.json; tail -3 $O/bench_cfg2.err
for n in 4 64; do timeout 300 python tools/gpu_large_bench.py cfg4 $n 3 2>&1 | grep -v Warn | head -3; done
for n in 16 64 128; do timeout 300 python tools/gpu_large_bench.py cfg2 $n 3 2>&1 | grep -v Warn | head -3; done
