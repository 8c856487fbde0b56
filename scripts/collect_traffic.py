#!/usr/bin/env python3
"""Turns rocprofv3 --pmc CSVs (one pass with FETCH_SIZE, one with WRITE_SIZE) into per-kernel HBM
traffic per launch, with the gfx950 correction for FETCH_SIZE.

    python scripts/collect_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> out.json

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half the bytes of wide coalesced
streaming reads (16 B per lane); kernels that stream with such loads get the x2 correction, the
others are reported as read (uncalibrated access widths), and both raw and corrected values are kept.
"""
import collections
import csv
import json
import sys

WIDE_STREAMING = ("fast_mask_kernel", "describe_tile_kernel")   # kernels whose global reads are 16 B per lane, coalesced


def per_kernel(path, counter):
    tot = collections.defaultdict(float)
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        tot[k] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    return {k: (tot[k], len(n[k])) for k in tot}


def short(name):
    name = name.split("(")[0]
    name = name.replace("void ", "").replace("orbamd::", "")
    return name.split("<")[0]


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k, (kib, launches) in fetch.items():
        s = short(k)
        f_raw = kib * 1024.0 / launches
        w = write.get(k, (0.0, 1))
        w_b = w[0] * 1024.0 / max(w[1], 1)
        corr = 2.0 if any(x in k for x in WIDE_STREAMING) else 1.0
        out[s] = {"launches": launches, "fetch_bytes_raw": f_raw, "fetch_correction": corr,
                  "write_bytes": w_b, "hbm_bytes_per_launch": f_raw * corr + w_b}
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for s, v in sorted(out.items()):
        print("%-24s %3d launches  fetch %.3e B (x%.0f)  write %.3e B  -> %.3e B/launch" %
              (s, v["launches"], v["fetch_bytes_raw"], v["fetch_correction"], v["write_bytes"], v["hbm_bytes_per_launch"]))


if __name__ == "__main__":
    main()
