O=gpurun_out/r03_w; mkdir -p $O; rm -f $O/modes2.txt
for k in 0 31 32 33; do echo "== fused up to (1,2,4,8 x CUs): stop $k" >> $O/modes2.txt
MULLS_DEBUG_STOP=$k MULLS_RESIDENT_MIN_PAIRS=100000 timeout 600 python tools/gpu_modes.py 128 192 256 384 512 1024 >> $O/modes2.txt 2>&1; done
cut -c1-12,126-200 $O/modes2.txt
