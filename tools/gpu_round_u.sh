cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r03_u; mkdir -p $O
rm -rf /tmp/prof_1
MULLS_SPLIT_MAX_PAIRS=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_1 -- python tools/gpu_icp_phases_lock.py 4096 > $O/run.txt 2>&1
python tools/kernel_stats.py /tmp/prof_1 "bench workload, 4096 pairs, lock-step" > $O/stats.txt; cat $O/stats.txt
for k in "void k_cert" k_nn_lds "void k_accum<1024"; do echo "== $k"; python tools/nn_series.py /tmp/prof_1 "$k" | tail -2; done
