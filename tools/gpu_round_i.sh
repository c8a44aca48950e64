O=gpurun_out/r03_i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_icp.py -m gpu -x -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log; tail -12 $O/pytest.log
timeout 600 python tools/gpu_modes.py 1024 4096 > $O/modes_wave.txt 2>&1; cat $O/modes_wave.txt
MULLS_WAVE_ACCUM_MIN_TRIPS=100000000 timeout 600 python tools/gpu_modes.py 1024 4096 > $O/modes_nowave.txt 2>&1; cat $O/modes_nowave.txt
