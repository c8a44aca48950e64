O=gpurun_out/r03_l; mkdir -p $O
timeout 900 python -m pytest tests/test_classify.py tests/test_gpu_map.py tests/test_gpu_adapter.py -m gpu -x -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log; tail -25 $O/pytest.log
timeout 300 python tools/gpu_odometry.py 12 > $O/odometry_resident.txt 2>&1; cat $O/odometry_resident.txt | tail -3
timeout 300 python tools/gpu_odometry.py 12 --host > $O/odometry_host.txt 2>&1; cat $O/odometry_host.txt | tail -3
