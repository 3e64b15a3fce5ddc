#!/usr/bin/env python3
"""Print a rocprofv3 --kernel-trace --stats kernel_stats.csv sorted by total time:  python tools/kstats.py <csv> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print("%9.3f ms  calls %5s  avg %9.2f us  %6s%%  %s" % (float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3,
                                                         r["Percentage"][:6], r["Name"][:90]))
