O=gpurun_out/r03_z; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_icp.py tests/test_demo_pair.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 300 python tools/gpu_modes.py 1 8 32 64 128 192 384
