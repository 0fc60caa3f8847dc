#!/usr/bin/env python
"""Print the head of a rocprofv3 *_kernel_stats.csv (share, calls, average duration), optionally only kernels whose name
contains one of the given substrings:  tools/kernel_stats_summary.py stats.csv [substring ...]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
keys = sys.argv[2:]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{sys.argv[1]}: {tot / 1e6:.1f} ms of kernel time")
for r in (rows if keys else rows[:25]):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:90]
    if keys and not any(k in n for k in keys):
        continue
    print(f"  {float(r['Percentage']):6.2f}% {int(r['Calls']):5d} calls avg {float(r['AverageNs']) / 1e3:8.1f} us  {n}")
