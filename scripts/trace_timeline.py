#!/usr/bin/env python3
"""Timeline of the kernels of single calls from a rocprofv3 kernel trace: dispatches separated by less than
`gap` microseconds form one call.  Prints the calls numbered first..last.
    python3 scripts/trace_timeline.py <kernel_trace.csv> [first] [last] [gap_us]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
last = int(sys.argv[3]) if len(sys.argv) > 3 else first + 1
gap = float(sys.argv[4]) if len(sys.argv) > 4 else 40.0
calls, cur, prev_end = [], [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev_end is not None and (s - prev_end) / 1e3 > gap:
        calls.append(cur)
        cur = []
    cur.append((r["Kernel_Name"].split("(")[0].replace("void ", "").replace("orbamd::", ""), s, e))
    prev_end = e
calls.append(cur)
for k in range(first, min(last + 1, len(calls))):
    c = calls[k]
    t0 = c[0][1]
    print("call %d: %d kernels, first start to last end %.1f us, sum of kernel durations %.1f us" %
          (k, len(c), (c[-1][2] - t0) / 1e3, sum(e - s for _, s, e in c) / 1e3))
    for name, s, e in c:
        print("   +%7.1f us  %6.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, name))
