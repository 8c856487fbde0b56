#!/usr/bin/env python3
"""Parity soak on the GPU box: many synthetic frames through the device-resident path, every
keypoint row, descriptor and match compared bit for bit with the CPU restatement (oracle/, run on a
thread pool).  Test infrastructure, not a benchmark.
    python scripts/soak_parity.py --preset 1080p --frames 1500 --first 100000
"""
import argparse
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="1080p", choices=["480p", "1080p", "4k"])
    ap.add_argument("--frames", type=int, default=600)
    ap.add_argument("--first", type=int, default=100000, help="index of the first synthetic frame")
    ap.add_argument("--batch", type=int, default=200)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--scene", default="sparse", choices=["sparse", "textured"],
                    help="sparse: the benchmark scene; textured: synth_textured_frame (about twice the keypoints)")
    ap.add_argument("--chunks", type=int, default=3, help="frame ranges pipelined on two streams (1 = off)")
    args = ap.parse_args()
    import torch
    import oracle
    from visual_slam_amd import synth
    from visual_slam_amd.engine import OrbEngine
    P = synth.PRESETS[args.preset]
    H, W = P["H"], P["W"]
    cap = {"480p": 4096, "1080p": 8192, "4k": 32768}[args.preset] * (2 if args.scene == "textured" else 1)
    eng = OrbEngine(0, cap=cap)
    oracle.c.lib()
    bad = 0
    nkp = ndesc = nmatch = 0
    done = 0
    with ThreadPoolExecutor(args.threads) as pool:
        while done < args.frames:
            B = min(args.batch, args.frames - done)
            if args.scene == "textured":
                gray = eng.synth_textured_frames(B, H, W, first_frame=args.first + done)
            else:
                gray = eng.synth_frames(B, H, W, first_frame=args.first + done, cell=P["cell"], density=P["density"])
            feat = eng.alloc_features(B)
            matches = eng.alloc_matches(B - 1)
            eng.extract_and_match(gray, feat, matches, chunks=args.chunks)
            torch.cuda.synchronize()
            host = gray.cpu().numpy()
            ex = list(pool.map(lambda i: oracle.c.extract_gray(host[i]), range(B)))
            mt = list(pool.map(lambda i: oracle.c.knn2(ex[i][1], ex[i - 1][1]), range(1, B)))
            counts = feat.counts.cpu().numpy()
            kp7 = feat.kp7.cpu().numpy()
            desc = feat.desc.cpu().numpy()
            vidx = feat.valid_idx.cpu().numpy()
            midx = matches.idx.cpu().numpy()
            mdist = matches.dist.cpu().numpy()
            for i in range(B):
                n, nv = int(counts[i, 0]), int(counts[i, 1])
                what = []
                if not (n == len(ex[i][0]) and nv == len(ex[i][1]) and n <= cap):
                    what.append("counts %d/%d vs %d/%d" % (n, nv, len(ex[i][0]), len(ex[i][1])))
                else:
                    if not np.array_equal(kp7[i, :n].view(np.uint32), ex[i][0].view(np.uint32)):
                        what.append("keypoint rows")
                    if not np.array_equal(desc[i, :nv], ex[i][1]):
                        what.append("descriptors (%d rows differ)" % int((desc[i, :nv] != ex[i][1]).any(axis=1).sum()))
                    if not np.array_equal(vidx[i, :nv], ex[i][2]):
                        what.append("valid_idx")
                    if i:
                        if not np.array_equal(midx[i - 1, :nv], mt[i - 1][0]):
                            what.append("match indices (%d rows differ)" % int((midx[i - 1, :nv] != mt[i - 1][0]).any(axis=1).sum()))
                        if not np.array_equal(mdist[i - 1, :nv], mt[i - 1][1]):
                            what.append("match distances")
                        nmatch += nv
                if what:
                    bad += 1
                    if bad <= 40:
                        print("MISMATCH at frame", args.first + done + i, "; ".join(what), flush=True)
                nkp += n
                ndesc += nv
            done += B
            print("%d / %d frames compared, %d keypoints, %d descriptors, %d match rows, %d bad frames"
                  % (done, args.frames, nkp, ndesc, nmatch, bad), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
