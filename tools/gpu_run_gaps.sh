#!/usr/bin/env bash
# Idle time of a resident batch run after run WITHOUT profiling: tools/gpu_run_gaps.sh <tag> <pairs>   (rocprofv3 kernel + memory-copy trace of tools/gpu_gaps_noprof.py;
# the last complete run: span to the next run's first kernel, time with at least one kernel or copy running, the largest gaps)
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-rungaps}; N=${2:-128}; O=$PWD/gpurun_out/$T; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python tools/gpu_gaps_noprof.py $N 8 > $O/run.log 2>&1
python - $O/trace <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:24]))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[:20]))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_clone_src")]
a, b = starts[-2], starts[-1]
# the run's first command may precede k_clone_src (fills, copies): walk back while gaps are small
step = rows[a:b]
span = (rows[b][0] - step[0][0]) / 1e3
end = step[0][0]; busy = 0.0; gaps = []; prev = step[0][2]
for s, e, n in step:
    if s > end:
        gaps.append(((s - end) / 1e3, prev, n)); busy += (e - s) / 1e3; end = e; prev = n
    elif e > end:
        busy += (e - end) / 1e3; end = e; prev = n
gaps.append(((rows[b][0] - end) / 1e3, prev, "next run's k_clone_src"))
print("run: %.0f us from its k_clone_src to the next run's, %.0f us with a kernel or copy running, %.0f us idle" % (span, busy, span - busy))
for g, p, n in sorted(gaps, reverse=True)[:10]:
    print("  gap %7.1f us between %-24s and %s" % (g, p, n))
PY
