#!/usr/bin/env python3
"""Developer tool: throughput when frames start in HOST memory (PCIe-inclusive rates).

  (a) the reference's call shape: one frame per call through the host-pointer C ABI
      (orbamd_extract_gray + orbamd_fetch, then orbamd_knn2 against the previous frame's descriptors);
  (b) batches copied from pinned host memory (one 64-frame chunk in flight while the previous one is
      processed) into the frame-batch path.

    python scripts/host_api_rate.py [--frames 256] [--preset 1080p]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402
from visual_slam_amd import synth  # noqa: E402
from visual_slam_amd.engine import OrbEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--preset", default="1080p")
    ap.add_argument("--chunk", type=int, default=64)
    a = ap.parse_args()
    P = synth.PRESETS[a.preset]
    eng = OrbEngine(0, cap=8192)
    gray = eng.synth_frames(a.frames, P["H"], P["W"], cell=P["cell"], density=P["density"])
    host = gray.cpu().numpy()

    # (a) single frames through the host-pointer API
    n = min(a.frames, 64)
    eng.ctx.extract_gray(host[0])
    t0 = time.perf_counter()
    prev = None
    for i in range(n):
        kp, desc, vidx = eng.ctx.extract_gray(host[i])
        if prev is not None:
            eng.ctx.knn2(desc, prev)
        prev = desc
    dt = time.perf_counter() - t0
    print("host-pointer API, one frame per call: %.1f frames/s (%d frames, extract + knn2 vs previous)" % (n / dt, n))

    # (b) pinned chunks, copy of chunk k+1 overlapped with processing of chunk k
    pinned = torch.from_numpy(host).pin_memory()
    copy_stream = torch.cuda.Stream()
    bufs = [torch.empty((a.chunk, P["H"], P["W"]), dtype=torch.uint8, device=eng.device) for _ in range(2)]
    ready = [torch.cuda.Event() for _ in range(2)]
    done = [torch.cuda.Event() for _ in range(2)]
    feat = eng.alloc_features(a.chunk)
    m = eng.alloc_matches(a.chunk - 1)

    def run():
        nchunks = a.frames // a.chunk
        for k in range(nchunks + 1):
            if k < nchunks:
                b = k & 1
                with torch.cuda.stream(copy_stream):
                    copy_stream.wait_event(done[b])
                    bufs[b].copy_(pinned[k * a.chunk:(k + 1) * a.chunk], non_blocking=True)
                    ready[b].record(copy_stream)
            if k > 0:
                b = (k - 1) & 1
                torch.cuda.current_stream().wait_event(ready[b])
                eng.extract_batch(bufs[b], out=feat)
                eng.match_consecutive(feat, out=m)
                done[b].record(torch.cuda.current_stream())
        torch.cuda.synchronize()

    for e in done:
        e.record(torch.cuda.current_stream())
    run()
    t0 = time.perf_counter()
    run()
    dt = time.perf_counter() - t0
    nf = (a.frames // a.chunk) * a.chunk
    print("pinned host -> device chunks of %d frames, copy overlapped: %.1f frames/s (%.1f GB/s of pixels)" %
          (a.chunk, nf / dt, nf * P["H"] * P["W"] / dt / 1e9))


if __name__ == "__main__":
    main()
