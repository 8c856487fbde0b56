"""The two CPU restatements (C, following the reference's loops; NumPy, closed-form) must agree bit
for bit, and both must reproduce the committed golden fixtures.  No GPU."""
import os

import numpy as np
import pytest

import oracle
from oracle import np_ref
from conftest import make_test_image
from visual_slam_amd import synth

import golden_util


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


CASES = [
    ("structured_120x160", lambda: make_test_image(120, 160, 21)),
    ("structured_odd_97x113", lambda: make_test_image(97, 113, 22)),
    ("noise_90x120", lambda: np.random.default_rng(23).integers(0, 256, (90, 120), dtype=np.uint8)),
    ("binary_80x100", lambda: (np.random.default_rng(24).integers(0, 2, (80, 100)) * 255).astype(np.uint8)),
    ("synth_small", lambda: synth.synth_frame(150, 200, 5, cell=12)),
    ("tiny_7x8", lambda: make_test_image(7, 8, 25, nrect=2)),
    ("tiny_5x5", lambda: make_test_image(5, 5, 26, nrect=2)),
]


@pytest.mark.parametrize("name,make", CASES, ids=[n for n, _ in CASES])
def test_extract_agrees(name, make):
    img = make()
    kp, desc, vidx = oracle.c.extract_gray(img)
    kp2, desc2, vidx2 = np_ref.extract_gray(img)
    assert np.array_equal(bits(kp), bits(kp2))
    assert np.array_equal(desc, desc2) and np.array_equal(vidx, vidx2)


def test_moments_agree():
    img = make_test_image(60, 70, 27)
    for x, y in [(0, 0), (69, 59), (35, 30), (3, 55), (68, 2), (-4, 20), (75, 64)]:
        assert tuple(oracle.c.moments(img, x, y)) == np_ref.moments(img, x, y) if 0 <= x < 70 and 0 <= y < 60 else True


def test_angle_from_moments_agrees():
    rng = np.random.default_rng(28)
    m = rng.integers(-1_900_000, 1_900_001, size=(20000, 2))
    m[:8] = [[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1], [1, 1], [-1, -1], [1, -1]]
    for m10, m01 in m:
        a = oracle.c.angle_from_moments(1.0, m10, m01)
        b = np_ref.angle_from_moments(1, int(m10), int(m01))
        assert a.view(np.uint32) == b.view(np.uint32)
        assert 0.0 <= float(a) <= 360.0


@pytest.mark.parametrize("nq,nt", [(1, 1), (7, 2), (40, 0), (130, 257), (500, 400)])
def test_knn2_agrees(nq, nt):
    q = synth.synth_descriptors(nq, 300 + nq, dup_frac=0.1)
    t = synth.synth_descriptors(nt, 400 + nt, dup_frac=0.1) if nt else np.zeros((0, 32), np.uint8)
    if nt > 10:
        t[nt - 1] = t[0]
        q[0] = t[0]
    idx, dist = oracle.c.knn2(q, t)
    idx2, dist2 = np_ref.knn2(q, t)
    assert np.array_equal(idx, idx2) and np.array_equal(dist, dist2)
    if nt >= 2:
        a = oracle.c.ratio_filter(idx, dist)
        b = np_ref.ratio_filter(idx, dist)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_bgr2gray_agrees():
    rng = np.random.default_rng(29)
    bgr = rng.integers(0, 256, (64, 77, 3), dtype=np.uint8)
    assert np.array_equal(oracle.c.bgr2gray(bgr), np_ref.bgr2gray(bgr))
    # pure colours: weights sum to 2^15, so white -> 255 and each channel alone rounds as published
    px = np.array([[[255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [0, 0, 0]]], np.uint8)
    assert oracle.c.bgr2gray(px).tolist() == [[255, 29, 150, 76, 0]]


def test_pattern_table_integrity():
    """Three copies of the sampling table exist, all generated from the reference's 1024 integers: the C
    restatement's own (reference order), the NumPy restatement's text file and the product's header (both
    pair-major).  Guard the content of each against the digests of the reference's integers."""
    import hashlib
    import re
    pts = oracle.c.pattern_points()
    assert pts.shape == (512, 2)
    assert hashlib.sha256(pts.reshape(-1).astype("<i4").tobytes()).hexdigest() == \
        "14a92c4fff1eeb6d126a65a063bc195ae5a86c259b4300eca1e6a5640eaf2c7c"
    assert hashlib.sha256(pts.reshape(-1).astype(np.int8).tobytes()).hexdigest() == \
        "96f37ade61dc2b1c02acea3a50df2c035bcf5a5b41bbb168e0550f36e2ad81e6"
    header = open(os.path.join(os.path.dirname(os.path.abspath(np_ref.__file__)), "..", "visual_slam_amd", "csrc",
                               "orb_pattern.h")).read()
    body = header[header.index("ORBAMD_PATTERN_QUADS_INIT {"):]
    product = np.array([int(v) for v in re.findall(r"-?\d+", body[body.index("{"):])], np.int64).reshape(256, 4)
    p = oracle.c.pattern_pairs()
    assert np.array_equal(product, p)
    assert p.shape == (256, 4) and np.abs(p).max() == 15
    assert hashlib.sha256(p.astype(np.int8).tobytes()).hexdigest() == \
        "55641b1c693e01a789bba85a9ea8c16d5ceca6fa858f2d69ad2a8781c0cd5dda"
    # back to the reference's order: points 0..255 are the first points, 256..511 the second
    flat = np.concatenate([p[:, 0:2].reshape(-1), p[:, 2:4].reshape(-1)]).astype("<i4")
    assert hashlib.sha256(flat.tobytes()).hexdigest() == \
        "14a92c4fff1eeb6d126a65a063bc195ae5a86c259b4300eca1e6a5640eaf2c7c"
    assert int(flat.sum()) == 21 and int(np.abs(flat).sum()) == 7755
    assert not ((p[:, 0] == 0) & (p[:, 1] == 0)).any() and not ((p[:, 2] == 0) & (p[:, 3] == 0)).any()
    assert np.array_equal(p, np.loadtxt(np_ref.__file__.replace("np_ref.py", "orb_pattern_pairs.txt"), dtype=np.int64))


def test_c_oracle_matches_golden():
    golden_util.check_impl_against_golden(oracle.c)


def test_numpy_restatement_matches_golden_small():
    class Impl:
        extract_gray = staticmethod(np_ref.extract_gray)
        knn2 = staticmethod(np_ref.knn2)
        ratio_filter = staticmethod(np_ref.ratio_filter)
        bgr2gray = staticmethod(np_ref.bgr2gray)
    golden_util.check_impl_against_golden(Impl, big=False)
