#!/usr/bin/env python3
"""print a rocprofv3 *_kernel_stats.csv compactly: name, calls, average us, percentage"""
import csv, sys, glob, os
for arg in sys.argv[1:]:
    fs = [arg] if os.path.isfile(arg) else glob.glob(os.path.join(arg, "**", "*kernel_stats.csv"), recursive=True)
    for f in fs:
        print("#", f)
        for r in csv.DictReader(open(f)):
            if float(r["Percentage"]) < 0.3:
                continue
            print(r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60].ljust(60), r["Calls"].rjust(5), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(9), r["Percentage"].rjust(6))
