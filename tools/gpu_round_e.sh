set -x
O=gpurun_out/r03_e; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log; tail -15 $O/pytest.log
timeout 600 python tools/gpu_modes.py 1 8 32 128 512 1024 > $O/modes.txt 2>&1; cat $O/modes.txt
MULLS_FEW_LAUNCHES_MAX_PAIRS=0 timeout 600 python tools/gpu_modes.py 1 8 32 128 512 > $O/modes_separate.txt 2>&1; cat $O/modes_separate.txt
