#!/usr/bin/env python3
"""timeline of the last N kernel dispatches of a rocprofv3 kernel trace: start (us, relative), duration, stream, name"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44]
    print("%9.1f %9.1f  dur %8.1f  q%s s%s  %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r["Queue_Id"], r["Stream_Id"], name))
