#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel name."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:28s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
