python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python tools/gpu_odometry.py 12 --check 3 2>&1 | tail -3
python tools/gpu_e2e_calls.py 8 1024 2>&1 | tail -4
python tools/gpu_modes.py 1 128 2>&1 | cut -c75-200
