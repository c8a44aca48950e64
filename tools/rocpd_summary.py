"""Turn a rocprofv3 rocpd SQLite database (--kernel-trace --stats) into the per-kernel text summary kept under profiles/."""
import sqlite3
import sys


def main(db_path, out_path, title):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = [title, "source: rocprofv3 --kernel-trace --stats (rocpd top_kernels view); durations in microseconds", "",
             "%-14s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
    for name, calls, total, avg, pct in rows:
        lines.append("%-14s %8d %14.1f %12.3f %7.2f%%" % (name.split("(")[0][:14], calls, total, avg, pct))
    open(out_path, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 kernel stats")
