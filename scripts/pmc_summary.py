#!/usr/bin/env python3
"""Per-kernel averages of every counter in a rocprofv3 counter_collection CSV.
    python3 scripts/pmc_summary.py <counter_collection.csv> [--json out.json] [kernel-name substring ...]
With --json the same figures are also written as {kernel: {counter: value per launch}} (merged into
the file if it exists, so that several passes end up in one file)."""
import csv
import json
import os
import sys
from collections import defaultdict

rows = csv.DictReader(open(sys.argv[1]))
want = sys.argv[2:]
json_out = None
if "--json" in want:
    i = want.index("--json")
    json_out = want[i + 1]
    del want[i:i + 2]
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

if json_out:
    data = json.load(open(json_out)) if os.path.exists(json_out) else {}
    for k in acc:
        short = k.replace("orbamd::", "")
        data.setdefault(short, {}).update({c: acc[k][c] / len(launches[k]) for c in acc[k]})
    json.dump(data, open(json_out, "w"), indent=1, sort_keys=True)
