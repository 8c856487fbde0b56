#!/usr/bin/env python3
"""Parity soak of the host-frame path on the GPU box: frames in host memory go through
HostFrameStager (pinned chunks, copy stream, matching across chunk seams); every result is compared
with the CPU restatement.  Test infrastructure.
    python scripts/soak_stager.py --frames 300 --chunk 32 [--bgr]
"""
import argparse
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--chunk", type=int, default=32)
    ap.add_argument("--bgr", action="store_true")
    ap.add_argument("--first", type=int, default=4242)
    args = ap.parse_args()
    import torch
    import oracle
    from visual_slam_amd import synth
    from visual_slam_amd.engine import OrbEngine
    from visual_slam_amd.staging import HostFrameStager
    P = synth.PRESETS["1080p"]
    H, W = P["H"], P["W"]
    eng = OrbEngine(0, cap=8192)
    dev = eng.synth_frames(args.frames, H, W, first_frame=args.first, cell=P["cell"], density=P["density"])
    torch.cuda.synchronize()
    gray = dev.cpu().numpy()
    del dev
    if args.bgr:
        rng = np.random.default_rng(1)
        frames = [np.stack([g, np.roll(g, 1, axis=1), 255 - g], axis=2) for g in gray]
        gray = np.stack([oracle.c.bgr2gray(f) for f in frames])
    else:
        frames = list(gray)
    oracle.c.lib()
    with ThreadPoolExecutor(16) as pool:
        ex = list(pool.map(lambda i: oracle.c.extract_gray(gray[i]), range(args.frames)))
        mt = [None] + list(pool.map(lambda i: oracle.c.knn2(ex[i][1], ex[i - 1][1]), range(1, args.frames)))
    st = HostFrameStager(eng, H, W, chunk=args.chunk, bgr=args.bgr)
    bad = seen = 0
    for ch in st.run(iter(frames)):
        # no explicit synchronisation: the copies below are ordered on the engine's stream
        counts = ch.features.counts.cpu().numpy()
        for j in range(ch.frames):
            i = ch.first + j
            n, nv = int(counts[j, 0]), int(counts[j, 1])
            ok = n == len(ex[i][0]) and nv == len(ex[i][1])
            ok = ok and np.array_equal(ch.features.kp7[j, :n].cpu().numpy().view(np.uint32), ex[i][0].view(np.uint32))
            ok = ok and np.array_equal(ch.features.desc[j, :nv].cpu().numpy(), ex[i][1])
            if i:
                ok = ok and np.array_equal(ch.matches.idx[j, :nv].cpu().numpy(), mt[i][0])
                ok = ok and np.array_equal(ch.matches.dist[j, :nv].cpu().numpy(), mt[i][1])
            if not ok:
                bad += 1
                if bad <= 10:
                    print("MISMATCH at frame", i, "(chunk first %d, slot %d)" % (ch.first, j), flush=True)
            seen += 1
    print("%d frames through the stager (chunk %d, %s), %d bad" % (seen, args.chunk, "bgr" if args.bgr else "gray", bad))
    sys.exit(1 if bad or seen != args.frames else 0)


if __name__ == "__main__":
    main()
