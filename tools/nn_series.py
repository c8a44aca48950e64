"""Durations of the search kernel launch by launch (one line per run of 20 iterations) from rocprofv3 --kernel-trace --output-format csv.
usage: nn_series.py <dir> [kernel-name-prefix]"""
import csv, glob, os, sys

pref = sys.argv[2] if len(sys.argv) > 2 else "k_nn_lds"
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Kernel_Name"].startswith(pref):
            rows.append((int(row["Start_Timestamp"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3))
rows.sort()
d = [r[1] for r in rows]
for i in range(0, len(d), 20):
    print(" ".join("%4.0f" % v for v in d[i:i + 20]), "| sum %.0f us" % sum(d[i:i + 20]))
