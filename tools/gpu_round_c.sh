set -x
mkdir -p gpurun_out/r03_c
timeout 600 python -m pytest tests/test_demo_pair.py tests/test_classify.py tests/test_ground_filter.py -m gpu -x -q > gpurun_out/r03_c/pytest.log 2>&1; echo "rc $?" >> gpurun_out/r03_c/pytest.log
tail -40 gpurun_out/r03_c/pytest.log
