set -x
mkdir -p gpurun_out/r03_b
timeout 600 python -m pytest tests/test_ground_filter.py -m gpu -x -q > gpurun_out/r03_b/pytest_ground.log 2>&1; echo "rc $?" >> gpurun_out/r03_b/pytest_ground.log
tail -30 gpurun_out/r03_b/pytest_ground.log
timeout 300 python tools/gpu_ground.py > gpurun_out/r03_b/ground.txt 2>&1; tail -20 gpurun_out/r03_b/ground.txt
