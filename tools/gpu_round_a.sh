set -x
mkdir -p gpurun_out/r03_a
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r03_a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03_a/pytest.log
timeout 120 tools/_bin/solve_bench 256 200 > gpurun_out/r03_a/solve_bench.txt 2>&1
timeout 600 python bench.py > gpurun_out/r03_a/bench.json 2> gpurun_out/r03_a/bench.err
timeout 600 python tools/gpu_modes.py 1 8 32 128 512 1024 4096 > gpurun_out/r03_a/modes.txt 2>&1
timeout 300 python tools/gpu_icp_phases.py 128 > gpurun_out/r03_a/icp_phases_128.txt 2>&1
timeout 300 python tools/gpu_icp_phases.py 1 > gpurun_out/r03_a/icp_phases_1.txt 2>&1
tail -3 gpurun_out/r03_a/pytest.log; cat gpurun_out/r03_a/solve_bench.txt; cat gpurun_out/r03_a/modes.txt; cat gpurun_out/r03_a/icp_phases_128.txt
