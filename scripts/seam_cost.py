#!/usr/bin/env python3
"""Cost of the seam work a rank > 0 adds to its step (pack of the boundary frame on the sender,
unpack + cross match on the receiver), measured on one GPU with the message looped back locally
(no transport)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visual_slam_amd import synth
from visual_slam_amd.engine import OrbEngine
from visual_slam_amd.scheduler import pack_boundary, unpack_boundary
P = synth.PRESETS["1080p"]
F = 1000
e = OrbEngine(0, cap=8192)
gray = e.synth_frames(F, P["H"], P["W"], cell=P["cell"], density=P["density"])
feat = e.alloc_features(F); m = e.alloc_matches(F - 1); seam = e.alloc_matches(1)
def step(with_seam):
    e.extract_and_match(gray, feat, m, chunks=4)
    if with_seam:
        msg = pack_boundary(feat, F - 1, True)
        b = unpack_boundary(msg, feat.cap, True)
        e.match_cross(feat, 0, b.desc, b.counts, out=seam)
for ws in (False, True, False, True):
    for _ in range(2): step(ws)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): step(ws)
    torch.cuda.synchronize()
    print("seam work %-5s %.3f ms per step" % (ws, (time.perf_counter() - t0) / 8 * 1e3))
