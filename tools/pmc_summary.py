"""Average rocprofv3 --pmc counter values per kernel launch. usage: pmc_summary.py <dir> [<dir> ...]"""
import csv, glob, os, sys
from collections import defaultdict

acc = defaultdict(float)
disp = defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            acc[(k, row["Counter_Name"])] += float(row["Counter_Value"])
            disp[(k, row["Counter_Name"])].add((f, row["Dispatch_Id"]))
kernels = sorted({k for k, _ in acc})
for k in kernels:
    parts = ["%s=%.4g" % (c, acc[(kk, c)] / len(disp[(kk, c)])) for kk, c in sorted(acc) if kk == k]
    n = max(len(disp[(kk, c)]) for kk, c in acc if kk == k)
    print("%-16s launches %4d  %s" % (k[:16], n, "  ".join(parts)))
