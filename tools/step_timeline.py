#!/usr/bin/env python3
"""Prints the kernel timeline of the last bench step from a rocprofv3 --kernel-trace CSV directory
(start, end, duration in microseconds relative to the step's first kernel)."""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_parse_cs" in r["Kernel_Name"]]
i0 = max(0, (starts[-1] if starts else 0) - 3)     # the last step: its decode and the fills before it
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f %8.1f %7.1f  gap %6.1f  q=%s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3,
                                                    r.get("Queue_Id", "?"), r["Kernel_Name"].replace("himut::", "")[:48]))
    prev_end = max(prev_end, e)
