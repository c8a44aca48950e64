O=gpurun_out/r03_n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_icp.py tests/test_demo_pair.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for mx in 0 1024 8192; do
echo "== SPLIT_MAX $mx" >> $O/modes.txt
MULLS_SPLIT_MAX_PAIRS=$mx timeout 600 python tools/gpu_modes.py 16 32 64 128 384 512 1024 2048 4096 >> $O/modes.txt 2>&1
done
cat $O/modes.txt
