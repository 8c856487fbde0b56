#!/usr/bin/env python3
"""Regenerates tests/golden/orb_golden.npz from the C restatement (oracle/orb_oracle.c).

The reference holds no golden vectors and cannot be run here, so these fixtures are produced by
this repo's own CPU restatement after it has passed the hand-derived known-answer tests and the
cross-check against the independent NumPy restatement.  Small cases are stored in full (inputs and
outputs); large cases store SHA-256 digests of the outputs (inputs are regenerated from seeds).

    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from conftest import make_test_image  # noqa: E402
from visual_slam_amd import synth  # noqa: E402


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.shape).encode())
        h.update(str(a.dtype).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def main():
    out = {}
    # small, stored in full
    img = make_test_image(72, 96, 101)
    kp, desc, vidx = oracle.c.extract_gray(img)
    out.update(small_img=img, small_kp=kp, small_desc=desc, small_vidx=vidx)
    q = synth.synth_descriptors(60, 7, dup_frac=0.1)
    t = synth.synth_descriptors(75, 8, dup_frac=0.1)
    t[40] = t[3]
    q[5] = t[3]
    idx, dist = oracle.c.knn2(q, t)
    gq, g2, g1 = oracle.c.ratio_filter(idx, dist)
    out.update(knn_q=q, knn_t=t, knn_idx=idx, knn_dist=dist, ratio_q=gq, ratio_t2=g2, ratio_t1=g1)
    rng = np.random.default_rng(55)
    bgr = rng.integers(0, 256, (31, 45, 3), dtype=np.uint8)
    out.update(bgr_in=bgr, bgr_gray=oracle.c.bgr2gray(bgr))
    # large, digests only
    digests = {}
    P = synth.PRESETS["480p"]
    frames = [synth.synth_frame(P["H"], P["W"], f, cell=P["cell"], density=P["density"]) for f in (0, 1)]
    ex = [oracle.c.extract_gray(f) for f in frames]
    digests["synth480p_frame0"] = digest(frames[0])
    digests["synth480p_extract0"] = digest(*ex[0])
    digests["synth480p_extract1"] = digest(*ex[1])
    digests["synth480p_knn_1_vs_0"] = digest(*oracle.c.knn2(ex[1][1], ex[0][1]))
    P = synth.PRESETS["1080p"]
    f1080 = synth.synth_frame(P["H"], P["W"], 3, cell=P["cell"], density=P["density"])
    digests["synth1080p_frame3"] = digest(f1080)
    digests["synth1080p_extract3"] = digest(*oracle.c.extract_gray(f1080))
    big = make_test_image(300, 420, 202)
    digests["structured_300x420_extract"] = digest(*oracle.c.extract_gray(big))
    out["digest_names"] = np.array(sorted(digests))
    out["digest_values"] = np.array([digests[k] for k in sorted(digests)])
    np.savez_compressed(os.path.join(HERE, "orb_golden.npz"), **out)
    for k in sorted(digests):
        print(k, digests[k])


if __name__ == "__main__":
    main()
