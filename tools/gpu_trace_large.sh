#!/usr/bin/env bash
# Kernel timeline of a large configuration (tools/gpu_large_bench.py): per-kernel table and the launch-by-launch series of the LAST run.
# usage: tools/gpu_trace_large.sh <tag> "<case pairs reps>" [lib]
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=$1; ARGS=$2; L=${3:-mulls_amd/libmulls_hip.so}; O=$PWD/gpurun_out/$T; mkdir -p $O
MULLS_HIP_LIB=$PWD/$L timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python tools/gpu_large_bench.py $ARGS > $O/run.log 2>&1
python tools/kernel_stats.py $O/trace "$T: $ARGS" > $O/stats.txt
python - $O/trace > $O/series.txt <<'PY'
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_clone_src")]
last = starts[-2] if len(starts) > 1 else starts[-1]  # the last unprofiled run (the very last one carries the library's events)
end = starts[-1] if len(starts) > 1 else len(rows)
ser = {}
for a, b, n in rows[last:end]:
    ser.setdefault(n, []).append((b - a) / 1e3)
span = (rows[end - 1][1] - rows[last][0]) / 1e3
busy = sum(b - a for a, b, n in rows[last:end]) / 1e3
print("one run: %.0f us from first kernel start to last kernel end, %.0f us inside kernels" % (span, busy))
for n, v in ser.items():
    print("%-22s n=%3d %s | sum %.0f us" % (n[:22], len(v), " ".join("%4.0f" % x for x in v[:42]), sum(v)))
PY
cat $O/stats.txt $O/series.txt
