import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from visual_slam_amd._lib import Context, MATCHER_VALU
rng = np.random.default_rng(1)
c = Context(0)
for nq, nt in ((5, 3), (40, 33), (64, 64), (100, 300), (300, 1000)):
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    c.set_matcher(0)
    i0, d0 = c.knn2(q, t)
    c.set_matcher(MATCHER_VALU)
    i1, d1 = c.knn2(q, t)
    bad = np.nonzero((i0 != i1).any(1) | (d0 != d1).any(1))[0]
    print(nq, nt, "mismatching queries:", len(bad))
    for b in bad[:6]:
        print("  q", b, "mfma", i0[b], d0[b], "valu", i1[b], d1[b])
