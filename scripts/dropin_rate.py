#!/usr/bin/env python3
"""Developer tool: microseconds per call of the single-frame, host-pointer entry points (the reference's call
shape: one pageable frame in, NumPy arrays out), with the library's own split of the host's time.

    python3 scripts/dropin_rate.py [--calls 60] [--preset 1080p]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_slam_amd import _lib, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--calls", type=int, default=60)
ap.add_argument("--preset", default="1080p")
args = ap.parse_args()
P = synth.PRESETS[args.preset]
H, W, K = P["H"], P["W"], args.calls
gray = [synth.synth_frame(H, W, i, cell=P["cell"], density=P["density"]) for i in range(9)]
bgr = [np.ascontiguousarray(np.repeat(g[..., None], 3, axis=2)) for g in gray]


def run(ctx, kind):
    frames = bgr if kind == "bgr" else gray
    ext = ctx.extract_bgr if kind == "bgr" else ctx.extract_gray
    _, dprev, _ = ext(frames[0])
    for i in range(3):
        ext(frames[1 + i])
    ctx.timing_enable(True)
    ctx.timing_reset()
    te, tm = [], []
    for i in range(K):
        f = frames[1 + i % 8]
        t0 = time.perf_counter()
        kp, d, v = ext(f)
        t1 = time.perf_counter()
        out = ctx.knn2_ratio(d, dprev)
        t2 = time.perf_counter()
        te.append(t1 - t0), tm.append(t2 - t1)
        dprev = d
    kt = ctx.timing_read()
    ht = ctx.host_timing_read()
    ctx.timing_enable(False)
    med = lambda v: np.median(v) * 1e6
    print("   extract %6.1f us   knn2+ratio %6.1f us   total %6.1f us = %.0f frames/s   keypoints %d" %
          (med(te), med(tm), med(te) + med(tm), 1e6 / (med(te) + med(tm)), len(kp)))
    print("   host shares per call (us):", {k: round(v / K, 1) for k, v in ht.items()},
          "  kernels:", {k: round(v[0] / K * 1e3, 1) for k, v in kt.items() if v[1]})


for staged in ("1", "0"):
    os.environ["ORBAMD_STAGE_COPIES"] = staged
    ctx = _lib.Context(0)
    for kind in ("gray", "bgr"):
        print("%s frame %dx%d, copies %s" % (kind, W, H, "staged through pinned memory" if staged == "1" else
                                             "handed to HIP as pageable arrays"))
        run(ctx, kind)
    ctx.close()
