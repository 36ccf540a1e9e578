#!/usr/bin/env python3
"""Sums a rocprofv3 counter_collection.csv per kernel (short name) and counter; prints one row per kernel."""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
names = []
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ah::", "").split("<")[0]
    c = r["Counter_Name"]
    acc[k][c] += float(r["Counter_Value"])
    if c not in names:
        names.append(c)
    calls[(k, c)] += 1
print("kernel,calls," + ",".join(names))
for k, d in acc.items():
    print(k + "," + str(calls[(k, names[0])]) + "," + ",".join("%.4g" % d.get(c, 0) for c in names))
