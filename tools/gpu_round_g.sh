O=gpurun_out/r03_g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_icp.py tests/test_gpu_stages.py tests/test_demo_pair.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 600 python tools/gpu_modes.py 1 8 32 128 256 512 1024 > $O/modes.txt 2>&1; cat $O/modes.txt
