#!/usr/bin/env python
"""Aggregate a rocprofv3 counter_collection.csv: per kernel, mean of each counter."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"][:60]
            if "conv_gemm" not in k and "wgrad_tr" not in k and "flash" not in k:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
