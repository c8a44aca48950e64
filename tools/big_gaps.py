"""Idle gaps above a threshold between consecutive GPU activities (kernels + copies) of a rocprofv3 --kernel-trace --memory-copy-trace csv, and kernels
longer than the threshold.  usage: big_gaps.py <dir> [threshold_us]"""
import csv, glob, sys
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 2000.0
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:24]))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[:18]))
rows.sort()
t0 = rows[0][0]
end = rows[0][1]
for i, (s, e, n) in enumerate(rows):
    if (s - end) / 1e3 > thr:
        print("gap %9.1f us at %10.1f ms  between %-24s and %-24s" % ((s - end) / 1e3, (s - t0) / 1e6, rows[i - 1][2], n))
    if (e - s) / 1e3 > thr:
        print("long %8.1f us at %10.1f ms  %s" % ((e - s) / 1e3, (s - t0) / 1e6, n))
    end = max(end, e)
print(len(rows), "activities over %.1f ms" % ((rows[-1][1] - t0) / 1e6))
