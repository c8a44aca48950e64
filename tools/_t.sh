python -m pytest tests/test_gpu_map.py tests/test_classify.py tests/test_motion_comp.py tests/test_gpu_adapter.py -x -q -m gpu 2>&1 | tail -3
python tools/gpu_odometry.py 12 --check 3 2>&1 | tail -3
python tools/gpu_odometry.py 12 --host 2>&1 | tail -2
python tools/gpu_map_bench.py 2>&1 | tail -4
