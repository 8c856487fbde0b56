#!/usr/bin/env python3
"""Developer tool: step time of OrbEngine.extract_and_match for several range counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visual_slam_amd import synth
from visual_slam_amd.engine import OrbEngine
P = synth.PRESETS["1080p"]
F = 1000
e = OrbEngine(0, cap=8192)
gray = e.synth_frames(F, P["H"], P["W"], cell=P["cell"], density=P["density"])
feat = e.alloc_features(F); m = e.alloc_matches(F - 1)
for c in (1, 2, 4, 6, 8, 12, 16):
    for _ in range(2):
        e.extract_and_match(gray, feat, m, chunks=c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        e.extract_and_match(gray, feat, m, chunks=c)
    torch.cuda.synchronize()
    print("ranges=%2d  %.3f ms per step" % (c, (time.perf_counter() - t0) / 5 * 1e3))
