#!/usr/bin/env python3
"""Prints the per-kernel averages of a rocprofv3 --kernel-trace --stats run (directory given)."""
import csv, glob, os, sys
hits = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True))
rows = list(csv.DictReader(open(hits[0])))
tot = 0.0
for r in rows:
    calls = int(r["Calls"])
    print("%-100s %5d %9.1f us" % (r["Name"][:100], calls, float(r["AverageNs"]) / 1e3))
