"""Independent NumPy restatement of the reference's ORB front-end.  TEST INFRASTRUCTURE ONLY.

Written from the reference's source text separately from ``orb_oracle.c`` and on purpose in a
different formulation (array-at-a-time, closed-form criteria instead of the reference's loops), so
that agreement between the two is evidence that both read the source the same way.

PARITY UNPINNED, like the C restatement: the reference holds no vectors for this path.

Scalar transcendental calls go through ``math`` (the C library's atan2/cos/sin, which is what the
reference calls); NumPy's own vectorised sin/cos may use different kernels and is avoided.
"""
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PAIRS = np.loadtxt(os.path.join(_HERE, "orb_pattern_pairs.txt"), dtype=np.int64).reshape(256, 4)

# Radius-3 circle, (dx, dy), k = 0 at the top, clockwise in image coordinates
# (reference src/orb.cpp:43-48).
_CIRCLE = [(0, -3), (1, -3), (2, -2), (3, -1), (3, 0), (3, 1), (2, 2), (1, 3),
           (0, 3), (-1, 3), (-2, 2), (-3, 1), (-3, 0), (-3, -1), (-2, -2), (-1, -3)]


def bgr2gray(bgr):
    """OpenCV 4.x 8-bit BGR2GRAY fixed-point formula (reference src/main_interface.cpp:83; unpinned)."""
    b = bgr[..., 0].astype(np.int64)
    g = bgr[..., 1].astype(np.int64)
    r = bgr[..., 2].astype(np.int64)
    return ((b * 3735 + g * 19235 + r * 9798 + 16384) >> 15).astype(np.uint8)


def detect_mask(gray):
    """Boolean corner mask, reference src/orb.cpp:51-102.

    The reference counts a run of >= 12 equal-class circle pixels while walking k = 0..15 once,
    so the run cannot wrap from k = 15 to k = 0; a corner therefore needs circle[s..s+11] all
    brighter (>= centre + 20) or all darker (<= centre - 20) for some s in 0..4.  The four-point
    pre-test (:65-68) is implied by that (k = 4 and 8 lie in every such run, and one of 0 / 12).
    """
    g = gray.astype(np.int32)
    H, W = g.shape
    mask = np.zeros((H, W), bool)
    if H < 7 or W < 7:
        return mask
    c = g[3:H - 3, 3:W - 3]
    br = []
    dk = []
    for dx, dy in _CIRCLE:
        n = g[3 + dy:H - 3 + dy, 3 + dx:W - 3 + dx]
        br.append(n >= c + 20)
        dk.append(n <= c - 20)
    br = np.stack(br)
    dk = np.stack(dk)
    hit = np.zeros(c.shape, bool)
    for s in range(5):
        hit |= br[s:s + 12].all(axis=0) | dk[s:s + 12].all(axis=0)
    mask[3:H - 3, 3:W - 3] = hit
    return mask


def detect_score_ext(gray, arc=12, wrap=False):
    """Corner score of the detector EXTENSION (not in the reference, whose detector has fixed
    parameters and no score; src/orb.cpp:39, :99-105 has the OpenCV call with a score commented out).

    score(p) = the largest t for which p passes the segment test with threshold t, inclusive
    compares as in the reference: the maximum over all runs of `arc` consecutive circle pixels
    (runs that pass k = 15 -> 0 only when `wrap`) of the smallest |difference| in the run, all
    brighter or all darker; 0 where no run is one-signed.  int32 [H, W], 0 in the 3-pixel border.
    With arc = 12, wrap = False: score >= 20  <=>  detect_mask()."""
    g = gray.astype(np.int32)
    H, W = g.shape
    out = np.zeros((H, W), np.int32)
    if H < 7 or W < 7:
        return out
    c = g[3:H - 3, 3:W - 3]
    d = np.stack([g[3 + dy:H - 3 + dy, 3 + dx:W - 3 + dx] - c for dx, dy in _CIRCLE])   # [16, h, w]
    best = np.zeros(c.shape, np.int32)
    starts = range(16) if wrap else range(16 - arc + 1)
    for s in starts:
        run = d[[(s + j) % 16 for j in range(arc)]]
        best = np.maximum(best, np.maximum(run.min(axis=0), (-run).min(axis=0)))
    out[3:H - 3, 3:W - 3] = np.maximum(best, 0)
    return out


def detect_ext(gray, threshold=20, arc=12, wrap=False, nms=False):
    """Detector extension: keypoints with score >= threshold; with `nms` only those whose score is
    strictly greater than the scores of all 8 neighbours (scores below the threshold count as 0).
    -> (xy int32 [N,2] in scan order, score int32 [N])."""
    sc = detect_score_ext(gray, arc, wrap)
    sc = np.where(sc >= threshold, sc, 0)
    keep = sc > 0
    if nms:
        p = np.pad(sc, 1)
        H, W = sc.shape
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dx or dy:
                    keep &= sc > p[1 + dy:1 + dy + H, 1 + dx:1 + dx + W]
    ys, xs = np.nonzero(keep)
    return np.stack([xs, ys], axis=1).astype(np.int32), sc[ys, xs].astype(np.int32)


def detect(gray):
    """-> xy int32 [N,2], ascending (y, x) = the reference's scan order (:51-52, :100)."""
    ys, xs = np.nonzero(detect_mask(gray))  # np.nonzero is row-major: y outer, x inner
    return np.stack([xs, ys], axis=1).astype(np.int32)


def moments(gray, x, y):
    """Raw moments over the clipped 31x31 square (reference src/orb.cpp:236-260), as integers."""
    H, W = gray.shape
    x0, x1 = max(x - 15, 0), min(x + 15, W - 1)
    y0, y1 = max(y - 15, 0), min(y + 15, H - 1)
    patch = gray[y0:y1 + 1, x0:x1 + 1].astype(np.int64)
    u = np.arange(x0, x1 + 1, dtype=np.int64) - x
    v = np.arange(y0, y1 + 1, dtype=np.int64) - y
    m00 = int(patch.sum())
    m10 = int((patch * u[None, :]).sum())
    m01 = int((patch * v[:, None]).sum())
    return m00, m10, m01


def angle_from_moments(m00, m10, m01):
    """Reference src/orb.cpp:124-138: double atan2, narrowed to float, +360 in double if negative."""
    if m00 == 0:
        return np.float32(0.0)
    a = np.float32(math.atan2(float(m01), float(m10)) * (180.0 / math.pi))
    if a < 0:
        a = np.float32(float(a) + 360.0)
    return a


def orient(gray, xy):
    return np.array([angle_from_moments(*moments(gray, int(x), int(y))) for x, y in xy], np.float32)


def describe_one(gray, x, y, angle_deg):
    """One 32-byte descriptor, reference src/orb.cpp:166-231."""
    H, W = gray.shape
    rad = float(np.float32(angle_deg)) * (math.pi / 180.0)
    c, s = math.cos(rad), math.sin(rad)
    px = _PAIRS[:, [0, 2]].astype(np.float64)  # x of point 1 / point 2
    py = _PAIRS[:, [1, 3]].astype(np.float64)
    # Two rounded products then one rounded sum/difference in double, narrowed to float.
    xr = ((px * c) - (py * s)).astype(np.float32)
    yr = ((px * s) + (py * c)).astype(np.float32)
    ix = np.rint(np.float32(x) + xr).astype(np.int64)  # float add, round half to even
    iy = np.rint(np.float32(y) + yr).astype(np.int64)
    inside = (ix >= 0) & (ix < W) & (iy >= 0) & (iy < H)
    val = np.where(inside, gray[np.clip(iy, 0, H - 1), np.clip(ix, 0, W - 1)], 0).astype(np.int64)
    bits = (val[:, 0] < val[:, 1]).astype(np.uint8)
    return np.packbits(bits, bitorder="little")  # bit k%8 of byte k//8


def describe(gray, xy, angle):
    """-> (desc [Nv,32], valid_idx [Nv]); 16-px border filter of reference src/orb.cpp:150-159."""
    H, W = gray.shape
    rows, vidx = [], []
    for i, (x, y) in enumerate(xy):
        if y - 16 >= 0 and y + 16 < H and x - 16 >= 0 and x + 16 < W:
            rows.append(describe_one(gray, int(x), int(y), angle[i]))
            vidx.append(i)
    desc = np.stack(rows) if rows else np.zeros((0, 32), np.uint8)
    return desc, np.array(vidx, np.int32)


def extract_gray(gray):
    xy = detect(gray)
    n = len(xy)
    kp = np.zeros((n, 7), np.float32)
    if n == 0:
        return kp, np.zeros((0, 32), np.uint8), np.zeros(0, np.int32)
    ang = orient(gray, xy)
    desc, vidx = describe(gray, xy, ang)
    kp[:, 0] = xy[:, 0]
    kp[:, 1] = xy[:, 1]
    kp[:, 2] = 7.0
    kp[:, 3] = ang
    kp[:, 6] = -1.0
    return kp, desc, vidx


_POP8 = np.array([bin(i).count("1") for i in range(256)], np.int32)


def hamming_matrix(q, t):
    q = np.asarray(q, np.uint8).reshape(-1, 32)
    t = np.asarray(t, np.uint8).reshape(-1, 32)
    return _POP8[q[:, None, :] ^ t[None, :, :]].sum(axis=2)


def knn2(q, t):
    """Two nearest train rows under (distance, index) order; -1 where the train set is too small."""
    d = hamming_matrix(q, t)
    nq, nt = d.shape
    idx = np.full((nq, 2), -1, np.int32)
    dist = np.full((nq, 2), -1, np.int32)
    if nt == 0 or nq == 0:
        return idx, dist
    order = np.argsort(d, axis=1, kind="stable")[:, :2]
    k = order.shape[1]
    idx[:, :k] = order
    dist[:, :k] = np.take_along_axis(d, order, axis=1)
    return idx, dist


def ratio_filter(idx, dist):
    """views.py:163-171: keep iff d1 < 0.75 * d2; returns (queryIdx, SECOND trainIdx, first trainIdx)."""
    if len(idx) and (idx < 0).any():
        raise ValueError("not enough values to unpack (expected 2)")
    keep = dist[:, 0].astype(np.float64) < dist[:, 1].astype(np.float64) * 0.75
    q = np.nonzero(keep)[0].astype(np.int32)
    return q, idx[keep, 1].astype(np.int32), idx[keep, 0].astype(np.int32)
