#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats --output-format csv: *_kernel_stats.csv -> the text table kept under profiles/."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("# rocprofv3 --kernel-trace --stats summary of %s" % sys.argv[1])
print("# durations in microseconds")
print("%-8s %-7s %-12s %-10s %-10s %-10s %s" % ("pct", "calls", "total_us", "avg_us", "min_us", "max_us", "kernel"))
for r in rows:
    print("%-8.2f %-7s %-12.1f %-10.2f %-10.2f %-10.2f %s" % (float(r["Percentage"]), r["Calls"], float(r["TotalDurationNs"]) / 1e3,
          float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Name"][:200]))
