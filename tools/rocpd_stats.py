#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace) as a per-kernel stats table
(the same columns as rocprofv3's kernel_stats.csv): name, calls, total/avg/min/max ns, percent."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
    for name, calls, tot, avg, mn, mx in rows:
        lines.append('"%s",%d,%d,%.1f,%d,%d,%.2f' % (name, calls, tot, avg, mn, mx, 100.0 * tot / total))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
