"""A stand-in for visual_slam_amd.engine.OrbEngine that computes with the CPU restatement, for
tests of host logic (the scheduler) on machines without a GPU.  TESTS ONLY."""
import numpy as np
import torch

import oracle
from visual_slam_amd.engine import BatchFeatures, BatchMatches


class OracleEngine:
    def __init__(self, cap=512):
        self.cap = cap

    def alloc_features(self, frames, cap=None):
        cap = cap or self.cap
        return BatchFeatures(kp7=torch.zeros((frames, cap, 7), dtype=torch.float32),
                             desc=torch.zeros((frames, cap, 32), dtype=torch.uint8),
                             valid_idx=torch.zeros((frames, cap), dtype=torch.int32),
                             counts=torch.zeros((frames, 2), dtype=torch.int32), cap=cap)

    def extract_batch(self, gray, out=None):
        frames = gray.shape[0]
        out = out if out is not None else self.alloc_features(frames)
        cap = out.cap
        kp7, desc, vidx, counts = out.kp7, out.desc, out.valid_idx, out.counts
        for f in range(frames):
            kp, d, v = oracle.c.extract_gray(gray[f].numpy())
            n, nv = min(len(kp), cap), int((v < cap).sum())
            kp7[f, :n] = torch.from_numpy(kp[:n])
            desc[f, :nv] = torch.from_numpy(d[:nv])
            vidx[f, :nv] = torch.from_numpy(v[:nv])
            # (n found, descriptor rows stored): the library's convention, also when n > cap
            counts[f, 0], counts[f, 1] = len(kp), int((v < cap).sum())
        return out

    def _match(self, q, t):
        idx, dist = oracle.c.knn2(q, t)
        good = np.zeros(len(q), np.uint8)
        if len(t) >= 2:
            good[oracle.c.ratio_filter(idx, dist)[0]] = 1
        return idx, dist, good

    def _pack(self, results, out=None):
        cap = self.cap
        m = out if out is not None else BatchMatches(idx=torch.full((max(len(results), 1), cap, 2), -7, dtype=torch.int32),
                         dist=torch.full((max(len(results), 1), cap, 2), -7, dtype=torch.int32),
                         good=torch.zeros((max(len(results), 1), cap), dtype=torch.uint8), cap=cap)
        for p, (idx, dist, good) in enumerate(results):
            m.idx[p, :len(idx)] = torch.from_numpy(idx)
            m.dist[p, :len(idx)] = torch.from_numpy(dist)
            m.good[p, :len(idx)] = torch.from_numpy(good)
        return m

    @staticmethod
    def _desc(feat_desc, counts, f, cap):
        nv = min(int(counts[f, 1]), cap)
        return feat_desc[f, :nv].numpy()

    def match_consecutive(self, feat, out=None):
        res = [self._match(self._desc(feat.desc, feat.counts, f, feat.cap), self._desc(feat.desc, feat.counts, f - 1, feat.cap))
               for f in range(1, feat.frames)]
        return self._pack(res, out)

    def match_cross(self, feat, qframe, t_desc, t_counts, tframe=0, out=None):
        q = self._desc(feat.desc, feat.counts, qframe, feat.cap)
        t = self._desc(t_desc, t_counts, tframe, feat.cap)
        return self._pack([self._match(q, t)], out)
