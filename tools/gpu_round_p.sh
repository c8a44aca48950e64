cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r03_p; mkdir -p $O
for f in 0 1; do
  rm -rf /tmp/prof_$f
  MULLS_FUSED_TGT_SETUP=$f MULLS_SPLIT_MAX_PAIRS=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$f -- python tools/gpu_one.py 3 4096 4 > /dev/null 2>&1
  python tools/kernel_stats.py /tmp/prof_$f "mode 3, 4096 pairs, fused target setup $f" > $O/stats_4096_fused$f.txt 2>&1
  cat $O/stats_4096_fused$f.txt
done
