# A/B of two builds on one box: per-step sums of k_cert and k_nn_lds (bench batch, 4096 pairs).  usage: gpu_ab_series.sh  (mulls_amd/libmulls_hip.so vs libmulls_hip_b.so)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
for L in libmulls_hip.so libmulls_hip_b.so libmulls_hip.so libmulls_hip_b.so; do
rm -rf /tmp/prof_1
MULLS_HIP_LIB=$PWD/mulls_amd/$L MULLS_SPLIT_MAX_PAIRS=0 timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_1 -- python tools/gpu_icp_phases_lock.py 4096 3 > /dev/null 2>&1
echo "$L: $(python tools/nn_series.py /tmp/prof_1 'void k_cert' | tail -1 | sed 's/.*|//') cert; $(python tools/nn_series.py /tmp/prof_1 k_nn_lds | tail -1 | sed 's/.*|//') nn"
done
