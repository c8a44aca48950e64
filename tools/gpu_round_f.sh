cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r03_f; mkdir -p $O
for cfg in "3 128" "3 1" "3 32"; do
  set -- $cfg
  rm -rf /tmp/prof_$1_$2
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$1_$2 -- python tools/gpu_one.py $1 $2 4 > /dev/null 2>&1
  python tools/kernel_gaps.py /tmp/prof_$1_$2 > $O/gaps_mode$1_$2pairs.txt 2>&1
  python tools/kernel_stats.py /tmp/prof_$1_$2 "mode $1, $2 pairs" > $O/stats_mode$1_$2pairs.txt 2>&1
done
tail -30 $O/gaps_mode3_128pairs.txt; tail -30 $O/gaps_mode3_1pairs.txt; cat $O/stats_mode3_128pairs.txt; cat $O/stats_mode3_1pairs.txt
