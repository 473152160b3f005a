"""Print / save the per-kernel summary of a rocprofv3 run (rocpd sqlite or kernel_stats csv)."""
import glob
import sqlite3
import sys


def main():
    d = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else None
    lines = []
    dbs = glob.glob(d + "/**/*.db", recursive=True)
    if dbs:
        c = sqlite3.connect(dbs[0])
        rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
        tot = sum(r[2] for r in rows)
        lines.append("# rocprofv3 --kernel-trace --stats summary (durations in us); total %.1f us" % tot)
        lines.append("%-110s %8s %12s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for r in rows[:40]:
            lines.append("%-110s %8d %12.1f %10.2f %7.2f" % (r[0][:110], r[1], r[2], r[3], r[4]))
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
