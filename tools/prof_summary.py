#!/usr/bin/env python3
"""Dump the per-kernel statistics of a rocprofv3 results database (rocpd sqlite, the default
output of `rocprofv3 --kernel-trace --stats`) as text: name, calls, total/avg/min/max duration."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
        "max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc"
    ).fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["# rocprofv3 --kernel-trace --stats summary of %s" % db,
             "# durations in microseconds",
             "%-8s %-7s %-12s %-10s %-10s %-10s %-5s %-5s %-7s %-8s %-10s %-5s %s"
             % ("pct", "calls", "total_us", "avg_us", "min_us", "max_us", "vgpr", "sgpr", "lds",
                "scratch", "grid_x", "wg_x", "kernel")]
    for r in rows:
        lines.append("%-8.2f %-7d %-12.1f %-10.2f %-10.2f %-10.2f %-5s %-5s %-7s %-8s %-10s %-5s %s"
                     % (100.0 * r[2] / total, r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3,
                        r[6], r[7], r[8], r[9], r[10], r[11],
                        r[0] if len(r[0]) <= 200 else r[0][:197] + "..."))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
