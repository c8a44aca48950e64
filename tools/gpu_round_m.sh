cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r03_m; mkdir -p $O
rm -rf /tmp/prof_odo
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_odo -- python tools/gpu_odometry.py 8 > $O/odo.txt 2>&1
python tools/kernel_stats.py /tmp/prof_odo "odometry, 8 frames" > $O/stats_odo.txt 2>&1
python - <<'PY' > gpurun_out/r03_m/frame_timeline.txt 2>&1
import csv, glob
rows=[]
for f in glob.glob("/tmp/prof_odo/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:22]))
for f in glob.glob("/tmp/prof_odo/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", ""))[:18]))
rows.sort()
# the last frame: find the last k_raw_mask (start of a frame's extraction)
starts=[i for i,r in enumerate(rows) if r[2].startswith("k_raw_mask")]
i0=starts[-1]
t0=rows[i0][0]; prev=t0
tk=0
for s,e,n in rows[i0:]:
    print("%8.1f us  +%6.1f gap  %-24s %7.1f us" % ((s-t0)/1e3, (s-prev)/1e3, n, (e-s)/1e3)); prev=e; tk+=(e-s)/1e3
print("frame span %.1f us, busy %.1f us, %d items" % ((rows[-1][1]-t0)/1e3, tk, len(rows)-i0))
PY
tail -3 $O/odo.txt; head -40 $O/stats_odo.txt; tail -130 $O/frame_timeline.txt
