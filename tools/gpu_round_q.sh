cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r03_q; mkdir -p $O
for k in 50 72 110; do
rm -rf /tmp/prof_1
MULLS_DEBUG_STOP=$k MULLS_SPLIT_MAX_PAIRS=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_1 -- python tools/gpu_one.py 3 4096 3 > /dev/null 2>&1
python tools/kernel_stats.py /tmp/prof_1 "x" > $O/stats_4096.txt 2>&1
echo "stop $k $(grep k_cert $O/stats_4096.txt)"
done
