"""Kernel durations and the idle gaps between consecutive kernels from a rocprofv3 --kernel-trace csv (last 70 launches).
usage: kernel_gaps.py <dir>"""
import csv, glob, sys
rows=[]
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
rows=rows[-70:]  # last run
prev=None
tot_k=0; tot_gap=0
for s,e,n in rows:
    gap = (s-prev)/1e3 if prev else 0
    print("%-18s dur %7.1f us  gap before %7.1f us" % (n[:18], (e-s)/1e3, gap))
    tot_k+=(e-s)/1e3; tot_gap+=gap
    prev=e
print("kernels", tot_k, "gaps", tot_gap)
