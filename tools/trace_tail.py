#!/usr/bin/env python3
"""dev tool: the last N kernel dispatches of a rocprofv3 kernel trace as a timeline (start / end in us, queue)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print("%8.1f %8.1f  q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r["Queue_Id"],
                                    r["Kernel_Name"].replace("void ed::tile::(anonymous namespace)::", "").replace("void ed::(anonymous namespace)::", "")[:60]))
