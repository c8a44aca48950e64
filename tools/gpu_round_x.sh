cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r03_x; mkdir -p $O
for n in 1 128; do
rm -rf /tmp/prof_$n
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$n -- python tools/gpu_one.py 3 $n 4 > /dev/null 2>&1
python tools/kernel_gaps.py /tmp/prof_$n > $O/gaps_$n.txt 2>&1
python tools/kernel_stats.py /tmp/prof_$n "mode 3, $n pairs" > $O/stats_$n.txt 2>&1
cat $O/stats_$n.txt; tail -22 $O/gaps_$n.txt
done
