#!/usr/bin/env python3
"""Parity fuzz on the GPU box: random frame sizes (rows of whole 16-byte groups and not), random content of
different corner densities, random batch lengths, through the device-resident batch path; every keypoint row,
descriptor, descriptor index and match compared bit for bit with the CPU restatement (oracle/).  Test
infrastructure, not a benchmark.
    python scripts/fuzz_parity.py --iters 300 --seed 1
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def random_frames(rng, B, H, W):
    """Frames whose corners move a little from one to the next, so that matching has structure."""
    kind = int(rng.integers(0, 4))
    base = np.full((H + 8, W + 8), int(rng.integers(60, 180)), np.int32)
    nrect = int(rng.integers(2, max(3, H * W // (200 if kind == 3 else 900))))
    for _ in range(nrect):
        h, w = int(rng.integers(1, 24)), int(rng.integers(1, 24))
        y, x = int(rng.integers(-4, H + 8)), int(rng.integers(-4, W + 8))
        base[max(y, 0):max(y + h, 0), max(x, 0):max(x + w, 0)] = int(rng.integers(0, 256))
    if kind == 1:
        base += rng.integers(-30, 31, size=base.shape)          # noise above the detector's threshold
    elif kind == 2:
        base[::int(rng.integers(2, 9))] = int(rng.integers(0, 256))    # stripes: runs along image rows
    frames = np.empty((B, H, W), np.uint8)
    for b in range(B):
        dy, dx = int(rng.integers(0, 5)), int(rng.integers(0, 5))
        f = base[dy:dy + H, dx:dx + W] + rng.integers(-3, 4, size=(H, W))
        frames[b] = np.clip(f, 0, 255).astype(np.uint8)
    return frames


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--single", action="store_true", help="also run each case's first frame through extract_gray")
    args = ap.parse_args()
    import torch
    import oracle
    from visual_slam_amd.engine import OrbEngine
    oracle.c.lib()
    rng = np.random.default_rng(args.seed)
    bad = 0
    nkp = ndesc = 0
    engines = {}
    for it in range(args.iters):
        H = int(rng.integers(40, 420))
        W = int(rng.integers(3, 40)) * 16 if rng.random() < 0.7 else int(rng.integers(40, 640))
        B = int(rng.integers(1, 7))
        host = random_frames(rng, B, H, W)
        ex = [oracle.c.extract_gray(host[i]) for i in range(B)]
        nmax = max(len(e[0]) for e in ex)
        cap = max(64, int(2 ** np.ceil(np.log2(nmax + 1))))
        eng = engines.get(cap)
        if eng is None:
            eng = engines[cap] = OrbEngine(0, cap=cap)
        gray = torch.from_numpy(host).cuda()
        feat = eng.extract_batch(gray)
        m = eng.match_consecutive(feat) if B > 1 else None
        eng.sync()
        what = []
        if args.single:                                   # the single-frame, host-pointer call on the first frame
            skp, sdesc, svidx = eng.ctx.extract_gray(host[0])
            if not (skp.shape == ex[0][0].shape and np.array_equal(skp.view(np.uint32), ex[0][0].view(np.uint32))
                    and np.array_equal(sdesc, ex[0][1]) and np.array_equal(svidx, ex[0][2])):
                what.append("single-frame call")
        for i in range(B):
            kp, desc, vidx = feat.frame(i)
            if kp.shape != ex[i][0].shape or not np.array_equal(kp.view(np.uint32), ex[i][0].view(np.uint32)):
                what.append("frame %d keypoint rows" % i)
            elif not np.array_equal(desc, ex[i][1]):
                what.append("frame %d descriptors" % i)
            elif not np.array_equal(vidx, ex[i][2]):
                what.append("frame %d descriptor indices" % i)
            elif i and len(ex[i][1]):
                eidx, edist = oracle.c.knn2(ex[i][1], ex[i - 1][1])
                idx, dist, _ = m.pair(i - 1, len(ex[i][1]))
                if not (np.array_equal(idx, eidx) and np.array_equal(dist, edist)):
                    what.append("pair %d matches" % i)
            nkp += len(ex[i][0])
            ndesc += len(ex[i][1])
        if what:
            bad += 1
            print("MISMATCH iteration %d (H=%d W=%d B=%d cap=%d): %s" % (it, H, W, B, cap, "; ".join(what)), flush=True)
        if (it + 1) % 25 == 0:
            print("%d / %d cases, %d keypoints, %d descriptors, %d bad" % (it + 1, args.iters, nkp, ndesc, bad), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
