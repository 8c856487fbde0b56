#!/usr/bin/env python3
"""Developer tool: do the extraction kernels (vector ALU) and the matcher (matrix cores) gain from
running at the same time?  Extraction of batch A on one stream and matching of batch B's features on
another, each alone and both together, on independent data (no dependency, no pipeline fill).

    python3 scripts/coexec_probe.py [--frames 1000] [--reps 5]

Workgroup residency can be steered with ORBAMD_PAD_LDS_DESCRIBE / ORBAMD_PAD_LDS_KNN2 (bytes of extra
LDS per workgroup) in the environment.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from visual_slam_amd import _lib, synth  # noqa: E402
from visual_slam_amd.engine import OrbEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=1000)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--preset", default="1080p")
args = ap.parse_args()
P = synth.PRESETS[args.preset]
F = args.frames
cap = {"480p": 4096, "1080p": 8192, "4k": 32768}[args.preset]

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
e1 = OrbEngine(0, cap=cap)
e2 = OrbEngine(0, cap=cap)
e1.bind_stream(s1)
e2.bind_stream(s2)
with torch.cuda.stream(s1):
    grayA = e1.synth_frames(F, P["H"], P["W"], first_frame=0, cell=P["cell"], density=P["density"])
    featA = e1.alloc_features(F)
with torch.cuda.stream(s2):
    grayB = e2.synth_frames(F, P["H"], P["W"], first_frame=5000, cell=P["cell"], density=P["density"])
    featB = e2.extract_batch(grayB)
    mB = e2.alloc_matches(F - 1)
torch.cuda.synchronize()


def run(what, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if what & 1:
            e1.extract_batch(grayA, out=featA)
        if what & 2:
            e2.match_consecutive(featB, out=mB)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for what in (1, 2, 3):
    run(what, 2)
res = {}
for name, what in (("extract alone", 1), ("match alone", 2), ("both, two streams", 3)):
    res[what] = min(run(what, args.reps) for _ in range(3))
    print("%-20s %.3f ms per step" % (name, res[what]))
print("sum of the parts %.3f ms, max %.3f ms, measured together %.3f ms -> overlap gain %.1f %% of the sum"
      % (res[1] + res[2], max(res[1], res[2]), res[3], 100.0 * (res[1] + res[2] - res[3]) / (res[1] + res[2])))
# per-kernel durations while both run (events on each stream)
for e in (e1, e2):
    e.ctx.timing_enable(True)
    e.ctx.timing_reset()
run(3, 3)
for tag, e in (("extract stream", e1), ("match stream", e2)):
    t = e.ctx.timing_read()
    print(tag, {k: round(v[0] / max(v[1], 1), 3) for k, v in t.items() if v[1]})
    e.ctx.timing_enable(False)
print("pads:", os.environ.get("ORBAMD_PAD_LDS_DESCRIBE", "0"), os.environ.get("ORBAMD_PAD_LDS_KNN2", "0"))
