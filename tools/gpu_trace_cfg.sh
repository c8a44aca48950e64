#!/usr/bin/env bash
# Kernel timeline of one bench configuration: per-kernel table and the launch-by-launch series of the search kernels of the LAST timed step.
# usage: tools/gpu_trace_cfg.sh <tag> "<bench.py arguments>" [lib]
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=$1; ARGS=$2; L=${3:-mulls_amd/libmulls_hip.so}; O=$PWD/gpurun_out/$T; mkdir -p $O
MULLS_HIP_LIB=$PWD/$L timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python bench.py $ARGS --no-other-configs --no-cpu-baseline --no-end-to-end --no-converging --sustain-s 0 > $O/run.log 2>&1
python tools/kernel_stats.py $O/trace "$T: $ARGS" > $O/stats.txt
python - $O/trace > $O/series.txt <<'PY'
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort()
# the last step: from the last k_clone_src on
last = max(i for i, r in enumerate(rows) if r[1].startswith("k_clone_src"))
ser = {}
for _, n, d in rows[last:]:
    ser.setdefault(n, []).append(d)
for n, v in ser.items():
    if len(v) >= 3:
        print("%-22s %s | sum %.0f us" % (n[:22], " ".join("%4.0f" % x for x in v[:44]), sum(v)))
    else:
        print("%-22s %s" % (n[:22], " ".join("%.0f" % x for x in v)))
PY
cat $O/stats.txt $O/series.txt
