O=gpurun_out/r03_h; mkdir -p $O
timeout 900 python tools/gpu_modes.py 1 4 8 32 64 128 192 256 384 512 1024 4096 > $O/modes.txt 2>&1; cat $O/modes.txt
