"""Per-kernel duration table from rocprofv3 --kernel-trace --output-format csv. usage: kernel_stats.py <dir> [title]"""
import csv, glob, os, sys
from collections import defaultdict

dur = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        dur[row["Kernel_Name"].split("(")[0]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
total = sum(sum(v) for v in dur.values())
if len(sys.argv) > 2:
    print(sys.argv[2])
print("%-18s %8s %12s %10s %8s %10s %10s" % ("kernel", "calls", "total_us", "avg_us", "pct", "min_us", "max_us"))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("%-18s %8d %12.1f %10.3f %7.2f%% %10.1f %10.1f" % (k[:18], len(v), sum(v), sum(v) / len(v), 100 * sum(v) / total, min(v), max(v)))
