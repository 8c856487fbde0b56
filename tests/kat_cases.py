"""Known-answer cases derived BY HAND from the reference's source text (no reference run needed).

Each case takes `impl`, any object with detect / orient / describe / extract_gray / knn2 /
ratio_filter in the calling convention of oracle.c_oracle, so the same cases pin the C
restatement, the NumPy restatement and the GPU library.  Expected values are worked out in the
comments from the cited reference lines.
"""
import numpy as np

# Radius-3 circle (dx, dy), k = 0 .. 15 (reference src/orb.cpp:43-48).
CIRCLE = [(0, -3), (1, -3), (2, -2), (3, -1), (3, 0), (3, 1), (2, 2), (1, 3),
          (0, 3), (-1, 3), (-2, 2), (-3, 1), (-3, 0), (-3, -1), (-2, -2), (-1, -3)]


def _has(xy, x, y):
    return bool(((xy[:, 0] == x) & (xy[:, 1] == y)).any())


def kat_single_bright_pixel(impl):
    """Flat 100 with one pixel 200 at (31, 30): that pixel sees 16 darker neighbours (100 <= 200-20),
    the run counter reaches 12 at k = 11 (src/orb.cpp:80-91).  Any other pixel sees at most one
    brighter neighbour.  Moments are symmetric -> m10 = m01 = 0 -> atan2(0, 0) = 0 (:127).  No pattern
    point is (0, 0), so every BRIEF test compares 100 < 100 -> all-zero descriptor (:220)."""
    g = np.full((64, 64), 100, np.uint8)
    g[30, 31] = 200
    kp, desc, vidx = impl.extract_gray(g)
    assert kp.tolist() == [[31.0, 30.0, 7.0, 0.0, 0.0, 0.0, -1.0]]
    assert desc.shape == (1, 32) and not desc.any()
    assert vidx.tolist() == [0]


def kat_no_wraparound(impl):
    """The run is counted for k = 0..15 once, never across k = 15 -> 0 (src/orb.cpp:70-96): a centre
    is a corner iff circle[s..s+11] is one class for some s in 0..4."""
    for s in range(0, 8):
        g = np.full((41, 41), 100, np.uint8)
        for k in range(s, s + 12):
            dx, dy = CIRCLE[k % 16]
            g[20 + dy, 20 + dx] = 160
        assert _has(impl.detect(g), 20, 20) == (s <= 4), "bright arc starting at k=%d" % s
    # 11 in a row is not enough; 13 is
    for run, want in ((11, False), (12, True), (13, True)):
        g = np.full((41, 41), 100, np.uint8)
        for k in range(2, 2 + run):
            dx, dy = CIRCLE[k]
            g[20 + dy, 20 + dx] = 30       # darker arc
        assert _has(impl.detect(g), 20, 20) == want


def kat_threshold_inclusive(impl):
    """Brighter means I_c >= I_p + 20, darker I_c <= I_p - 20 (src/orb.cpp:75, :80), in int
    arithmetic: no wrap at the ends of the 8-bit range."""
    def centre_is_corner(p, c):
        g = np.full((41, 41), p, np.uint8)
        for dx, dy in CIRCLE:
            g[20 + dy, 20 + dx] = c
        return _has(impl.detect(g), 20, 20)
    assert centre_is_corner(100, 120) and not centre_is_corner(100, 119)
    assert centre_is_corner(100, 80) and not centre_is_corner(100, 81)
    assert centre_is_corner(235, 255) and not centre_is_corner(236, 255) and not centre_is_corner(250, 255)
    assert centre_is_corner(20, 0) and not centre_is_corner(19, 0) and not centre_is_corner(5, 0)
    assert not centre_is_corner(255, 255) and not centre_is_corner(0, 0)


def kat_border_keypoints_have_no_descriptor(impl):
    """detectKeypoints scans [3, W-3) x [3, H-3) (src/orb.cpp:51-52) but computeDescriptors keeps only
    x-16 >= 0, x+16 < W, y-16 >= 0, y+16 < H (:156): border keypoints stay in the keypoint array and
    get no descriptor row, so rows of the two arrays do not line up."""
    H, W = 60, 70
    g = np.full((H, W), 100, np.uint8)
    pts = [(3, 3), (15, 30), (16, 30), (30, 15), (30, 16), (W - 17, 30), (W - 16, 30), (W - 4, 30),
           (30, H - 17), (30, H - 16), (W - 4, H - 4), (40, 43)]
    for x, y in pts:
        g[y, x] = 220
    # pixels outside the scan range never become keypoints even when isolated and bright
    g[2, 40] = 220
    g[30, W - 3] = 220
    kp, desc, vidx = impl.extract_gray(g)
    exp = sorted(pts, key=lambda p: (p[1], p[0]))            # scan order: y outer, x inner
    assert [(int(r[0]), int(r[1])) for r in kp] == exp
    want_valid = [i for i, (x, y) in enumerate(exp) if 16 <= x <= W - 17 and 16 <= y <= H - 17]
    assert vidx.tolist() == want_valid
    assert desc.shape == (len(want_valid), 32)


def kat_orientation_axes(impl):
    """One lit pixel at offset (u, v) from the keypoint on a black image: m10 = u*I, m01 = v*I, so
    angle = atan2(v, u) in degrees mapped to [0, 360) (src/orb.cpp:127-132)."""
    cases = {(1, 0): 0.0, (0, 1): 90.0, (-1, 0): 180.0, (0, -1): 270.0, (5, 5): 45.0, (-7, 7): 135.0,
             (-3, -3): 225.0, (15, -15): 315.0}
    for (u, v), want in cases.items():
        g = np.zeros((50, 50), np.uint8)
        g[25 + v, 25 + u] = 200
        ang = impl.orient(g, np.array([[25, 25]], np.int32))
        assert ang.dtype == np.float32 and float(ang[0]) == want, ((u, v), float(ang[0]))
    # all-black patch: m00 == 0 -> angle 0 (:134-138)
    assert float(impl.orient(np.zeros((50, 50), np.uint8), np.array([[25, 25]], np.int32))[0]) == 0.0
    # a pixel outside the 31x31 square does not count (u = 16)
    g = np.zeros((50, 50), np.uint8)
    g[25, 41] = 200
    assert float(impl.orient(g, np.array([[25, 25]], np.int32))[0]) == 0.0
    # clipping at the image corner (:250): uniform image, keypoint (0, 0) sees only u, v in [0, 15]
    # -> m10 = m01 > 0 -> 45 degrees
    g = np.full((40, 40), 100, np.uint8)
    assert float(impl.orient(g, np.array([[0, 0]], np.int32))[0]) == 45.0


def _pairs():
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    return np.loadtxt(os.path.join(here, "..", "oracle", "orb_pattern_pairs.txt"), dtype=np.int64).reshape(256, 4)


def kat_descriptor_single_lit_sample(impl):
    """Black image, one lit pixel at centre + p2[k0] (angle 0) or at centre + (-y, x) of p2[k0] =
    (x, y) (angle 90): bit k is set iff pixel(p1[k]) < pixel(p2[k]) (src/orb.cpp:220), i.e. exactly
    for the pairs whose second point lands on the lit pixel and whose first does not."""
    P = _pairs()
    cx = cy = 40
    for k0 in (0, 17, 100, 255):
        x, y = int(P[k0, 2]), int(P[k0, 3])
        for angle, (sx, sy) in ((0.0, (x, y)), (90.0, (-y, x)), (180.0, (-x, -y)), (270.0, (y, -x))):
            g = np.zeros((81, 81), np.uint8)
            g[cy + sy, cx + sx] = 255
            desc, vidx = impl.describe(g, np.array([[cx, cy]], np.int32), np.array([angle], np.float32))
            want_bits = np.array([(P[k, 2] == x and P[k, 3] == y) and not (P[k, 0] == x and P[k, 1] == y)
                                  for k in range(256)], np.uint8)
            assert want_bits[k0] == 1
            want = np.packbits(want_bits, bitorder="little")
            assert vidx.tolist() == [0]
            assert np.array_equal(desc[0], want), (k0, angle)


def kat_descriptor_out_of_image_reads_zero(impl):
    """Samples that fall outside the image read 0 (src/orb.cpp:211-217).  On a uniform non-zero image
    with the keypoint exactly 16 px from the top-left corner, a pair sets its bit iff its first
    point leaves the image and its second does not."""
    P = _pairs()
    H = W = 48
    g = np.full((H, W), 50, np.uint8)
    cx = cy = 16
    desc, vidx = impl.describe(g, np.array([[cx, cy]], np.int32), np.array([0.0], np.float32))
    inside = lambda x, y: 0 <= cx + x < W and 0 <= cy + y < H
    want_bits = np.array([(not inside(P[k, 0], P[k, 1])) and inside(P[k, 2], P[k, 3]) for k in range(256)], np.uint8)
    assert not want_bits.any()      # all pattern points lie within +-15, so at angle 0 nothing leaves
    assert np.array_equal(desc[0], np.packbits(want_bits, bitorder="little"))
    # rotated by 45 degrees the far points (radius up to 21.2) do leave the image
    desc45, _ = impl.describe(g, np.array([[cx, cy]], np.int32), np.array([45.0], np.float32))
    import math
    c, s = math.cos(45.0 * (math.pi / 180.0)), math.sin(45.0 * (math.pi / 180.0))
    def rot_inside(x, y):
        xr = np.float32(x * c - y * s)
        yr = np.float32(x * s + y * c)
        px = int(np.rint(np.float32(cx) + xr))
        py = int(np.rint(np.float32(cy) + yr))
        return 0 <= px < W and 0 <= py < H
    want45 = np.array([(not rot_inside(P[k, 0], P[k, 1])) and rot_inside(P[k, 2], P[k, 3]) for k in range(256)], np.uint8)
    assert want45.any()
    assert np.array_equal(desc45[0], np.packbits(want45, bitorder="little"))


def kat_matcher_ties(impl):
    """Two smallest under (distance, train index) order; equal distances keep ascending index;
    fewer than two train rows leave -1 (OpenCV batchDistance insertion with strict '<')."""
    z = np.zeros((1, 32), np.uint8)
    t = np.zeros((5, 32), np.uint8)
    t[0, 0] = 0x07          # distance 3
    t[1, 0] = 0x01          # distance 1
    t[2, 5] = 0x80          # distance 1 (tie with row 1 -> row 1 first)
    t[3, 0] = 0x01          # distance 1 (later duplicate of row 1: never reported)
    t[4, :] = 0xFF          # distance 256
    idx, dist = impl.knn2(z, t)
    assert idx.tolist() == [[1, 2]] and dist.tolist() == [[1, 1]]
    idx, dist = impl.knn2(z, t[4:5])
    assert idx.tolist() == [[0, -1]] and dist.tolist() == [[256, -1]]
    idx, dist = impl.knn2(z, t[:0])
    assert idx.tolist() == [[-1, -1]] and dist.tolist() == [[-1, -1]]
    idx, dist = impl.knn2(np.repeat(z, 3, axis=0), t[[0, 4]])
    assert idx.tolist() == [[0, 1]] * 3 and dist.tolist() == [[3, 256]] * 3


def kat_ratio_boundary(impl):
    """Kept iff d1 < 0.75 * d2 (views.py:168): (3, 4) is not kept (3 < 3.0 is false), (2, 3) is.
    The second stored index is m2.trainIdx (views.py:170)."""
    idx = np.array([[10, 11], [20, 21], [30, 31], [40, 41]], np.int32)
    dist = np.array([[3, 4], [2, 3], [0, 0], [0, 1]], np.int32)
    gq, g2, g1 = impl.ratio_filter(idx, dist)
    assert gq.tolist() == [1, 3] and g2.tolist() == [21, 41] and g1.tolist() == [20, 40]
    try:
        impl.ratio_filter(np.array([[0, -1]], np.int32), np.array([[7, -1]], np.int32))
    except ValueError:
        pass
    else:
        raise AssertionError("a query with one neighbour must raise, as tuple unpacking does (views.py:167)")


ALL = [kat_single_bright_pixel, kat_no_wraparound, kat_threshold_inclusive,
       kat_border_keypoints_have_no_descriptor, kat_orientation_axes, kat_descriptor_single_lit_sample,
       kat_descriptor_out_of_image_reads_zero, kat_matcher_ties, kat_ratio_boundary]
