#!/usr/bin/env bash
# The round's last call on the final build: the whole GPU suite, smoke(), and the differential sweeps beyond the suite (default options): tools/gpu_closing_checks.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-closing}; mkdir -p $O
{
echo "== python -m pytest tests -m gpu -q"; timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
echo "== __graft_entry__.smoke()"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== python tools/gpu_fuzz_mixed.py 60 16"; timeout 300 python tools/gpu_fuzz_mixed.py 60 16 2>&1 | tail -2
echo "== python tools/gpu_fuzz_batch.py 8"; timeout 300 python tools/gpu_fuzz_batch.py 8 2>&1 | tail -2
echo "== python tools/gpu_fuzz_sweep.py 12"; timeout 300 python tools/gpu_fuzz_sweep.py 12 2>&1 | tail -2
echo "== python tools/gpu_odometry.py 12 --check 3"; timeout 120 python tools/gpu_odometry.py 12 --check 3 2>&1 | tail -2
} > $O/closing.txt 2>&1
cat $O/closing.txt
