#!/usr/bin/env bash
# Where a headline step's time goes besides its kernels: the last step of a short bench run under rocprofv3 (kernel + memory-copy trace): span, busy time, the gaps of
# more than 3 us with the kernels on either side.   tools/gpu_step_gaps.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-gaps}; O=$PWD/gpurun_out/$T; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python bench.py --no-other-configs --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end --no-converging --sustain-s 0 > $O/run.log 2>&1
python - $O/trace <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:26]))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[:20]))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_clone_src")]
a, b = starts[-2], starts[-1]
step = rows[a:b]
span = (rows[b][0] - step[0][0]) / 1e3
busy = 0.0; end = step[0][0]
gaps = []
for s, e, n in step:
    if s > end:
        gaps.append(((s - end) / 1e3, prev, n))
    busy += (max(e, end) - max(s, end)) / 1e3 if e > end else 0.0
    if e > end: end = e; prev = n
gaps.append(((rows[b][0] - end) / 1e3, prev, "next step's k_clone_src"))
print("step: %.0f us from its first kernel to the next step's first kernel, %.0f us with a kernel or copy running, %.0f us idle" % (span, busy, span - busy))
for g, p, n in sorted(gaps, reverse=True)[:12]:
    print("  gap %7.1f us between %-26s and %s" % (g, p, n))
print("  gaps of 1-3 us: %d, sum %.0f us" % (sum(1 for g in gaps if 1 <= g[0] < 3), sum(g[0] for g in gaps if 1 <= g[0] < 3)))
PY
