cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r03_q; mkdir -p $O
for lib in a b a b; do
rm -rf /tmp/prof_1
L=mulls_amd/libmulls_hip.so; [ $lib = b ] && L=mulls_amd/libmulls_hip_b.so
MULLS_HIP_LIB=$PWD/$L MULLS_SPLIT_MAX_PAIRS=0 timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_1 -- python tools/gpu_icp_phases_lock.py 4096 3 > /dev/null 2>&1
python tools/kernel_stats.py /tmp/prof_1 "x" > $O/stats_4096.txt 2>&1
echo "$lib $(grep k_cert $O/stats_4096.txt)"
done
rm -rf /tmp/pmc_w
MULLS_SPLIT_MAX_PAIRS=0 timeout 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -- python tools/gpu_icp_phases_lock.py 4096 2 > /dev/null 2>&1
python tools/pmc_summary.py /tmp/pmc_w | grep k_cert
