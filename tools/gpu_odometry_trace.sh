#!/usr/bin/env bash
# Kernel timeline of the odometry front end's last frames: tools/gpu_odometry_trace.sh <tag> [launches]   (durations + gaps; rocprofv3 --kernel-trace)
set -u
TAG=${1:-odo_trace}; N=${2:-260}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- python tools/gpu_odometry.py 8 > $OUT/run.log 2>&1
python - "$OUT/trace" $N > $OUT/timeline.txt <<'PY'
import csv, glob, sys
rows=[]
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:28]))
for f in glob.glob(sys.argv[1]+"/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY "+r.get("Direction","")[:22]))
rows.sort()
rows=rows[-int(sys.argv[2]):]
t0=rows[0][0]; prev=None
for s,e,n in rows:
    gap=(s-prev)/1e3 if prev else 0
    print("%9.1f us  %-30s dur %7.1f  gap %7.1f%s" % ((s-t0)/1e3, n, (e-s)/1e3, gap, "   <<<<" if gap>30 else ""))
    prev=max(prev or 0,e)
PY
tail -5 $OUT/run.log; wc -l $OUT/timeline.txt
