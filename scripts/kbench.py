#!/usr/bin/env python3
"""Developer tool: per-kernel timings (HIP events) of the frame-batch path on synthetic frames.

    python scripts/kbench.py [--frames 200] [--preset 1080p] [--reps 3]

Prints milliseconds per launch for every kernel and the same scaled to 1000 frames.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
from visual_slam_amd import synth  # noqa: E402
from visual_slam_amd.engine import OrbEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--preset", default="1080p")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--cap", type=int, default=8192)
    ap.add_argument("--scene", default="sparse", choices=["sparse", "textured"])
    ap.add_argument("--detector", default="", help="detector extension as threshold,arc,wrap,nms (e.g. 20,9,1,1)")
    a = ap.parse_args()
    P = synth.PRESETS[a.preset]
    eng = OrbEngine(0, cap=a.cap)
    if a.detector:
        thr, arc, wrap, nms = (int(v) for v in a.detector.split(","))
        eng.ctx.set_detector(thr, arc, bool(wrap), bool(nms))
    if a.scene == "textured":
        gray = eng.synth_textured_frames(a.frames, P["H"], P["W"])
    else:
        gray = eng.synth_frames(a.frames, P["H"], P["W"], cell=P["cell"], density=P["density"])
    feat = eng.alloc_features(a.frames)
    m = eng.alloc_matches(a.frames - 1)
    for _ in range(2):
        eng.extract_batch(gray, out=feat)
        eng.match_consecutive(feat, out=m)
    eng.sync()
    eng.ctx.timing_enable(True)
    eng.ctx.timing_reset()
    for _ in range(a.reps):
        eng.extract_batch(gray, out=feat)
        eng.match_consecutive(feat, out=m)
    t = eng.ctx.timing_read()
    tot = 0.0
    for k, (ms, n) in t.items():
        if n:
            per = ms / n
            tot += per
            print("%-16s %9.4f ms/launch  %9.3f ms per 1000 frames" % (k, per, per * 1000.0 / a.frames))
    print("%-16s %9.4f ms           %9.3f ms per 1000 frames" % ("total", tot, tot * 1000.0 / a.frames))
    c = feat.counts.float().mean(0).tolist()
    print("mean keypoints %.0f, descriptors %.0f" % (c[0], c[1]))


if __name__ == "__main__":
    main()
