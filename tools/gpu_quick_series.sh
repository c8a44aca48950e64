cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf /tmp/prof_1
MULLS_SPLIT_MAX_PAIRS=0 timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_1 -- python tools/gpu_icp_phases_lock.py 4096 3 > /dev/null 2>&1
python tools/kernel_stats.py /tmp/prof_1 "bench batch, 4096 pairs" | head -8
for k in "void k_cert" k_nn_lds; do echo "== $k"; python tools/nn_series.py /tmp/prof_1 "$k" | tail -1; done
