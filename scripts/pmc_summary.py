#!/usr/bin/env python3
"""Per-kernel averages of every counter in a rocprofv3 counter_collection CSV.
    python3 scripts/pmc_summary.py <counter_collection.csv> [kernel-name substring ...]"""
import csv
import sys
from collections import defaultdict

rows = csv.DictReader(open(sys.argv[1]))
want = sys.argv[2:]
acc = defaultdict(lambda: defaultdict(float))
launches = defaultdict(set)
for r in rows:
    k = r["Kernel_Name"].split("(")[0].split("<")[0].strip()
    k = k.replace("void ", "")
    if want and not any(w in k for w in want):
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    launches[k].add(r["Dispatch_Id"])
for k in sorted(acc):
    n = len(launches[k])
    print("%-28s %d launches" % (k, n))
    for c in sorted(acc[k]):
        print("    %-24s %.4e per launch" % (c, acc[k][c] / n))
