"""GPU: the drop-in surfaces -- pybind11 module `build.orb_project` (the reference's module name and
location) and the Python host mirror of the reference's interface."""
import importlib

import numpy as np
import pytest

import oracle
from conftest import make_test_image
from visual_slam_amd import synth

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _bgr(seed, H=150, W=210):
    rng = np.random.default_rng(seed)
    g = make_test_image(H, W, seed)
    bgr = np.stack([g, np.roll(g, 2, axis=0), g[::-1, ::-1]], axis=2).astype(np.int32)
    return (bgr + rng.integers(-10, 11, bgr.shape)).clip(0, 255).astype(np.uint8)


def test_module_extract_is_the_reference_contract():
    import build.orb_project as orb
    img = _bgr(1)
    kp, desc = orb.extract(img)
    ekp, edesc = oracle.c.extract(img)
    assert kp.dtype == np.float32 and kp.ndim == 2 and kp.shape[1] == 7
    assert desc.dtype == np.uint8 and desc.shape[1] == 32
    assert np.array_equal(bits(kp), bits(ekp)) and np.array_equal(desc, edesc)
    assert kp.flags["OWNDATA"] or kp.base is not None
    # keyword name `image`, as the reference binds it (src/main_interface.cpp:107-109)
    kp2, desc2 = orb.extract(image=img)
    assert np.array_equal(bits(kp2), bits(kp)) and np.array_equal(desc2, desc)
    # non-contiguous input is handled by value (the reference would misread it)
    wide = np.zeros((150, 420, 3), np.uint8)
    wide[:, ::2] = img
    kp3, desc3 = orb.extract(wide[:, ::2])
    assert np.array_equal(bits(kp3), bits(kp)) and np.array_equal(desc3, desc)


def test_module_extract_without_keypoints_returns_0x0_descriptors():
    import build.orb_project as orb
    kp, desc = orb.extract(np.full((64, 80, 3), 90, np.uint8))
    assert kp.shape == (0, 7) and kp.dtype == np.float32
    assert desc.shape == (0, 0) and desc.dtype == np.uint8
    # keypoints only in the border band: N > 0, Nv = 0 -> [0, 32]
    img = np.full((64, 80, 3), 90, np.uint8)
    img[5, 40] = 250
    kp, desc = orb.extract(img)
    assert kp.shape == (1, 7) and desc.shape == (0, 32)


def test_module_rejects_what_the_reference_rejects():
    import build.orb_project as orb
    with pytest.raises(RuntimeError):
        orb.extract(np.zeros((10, 10, 4), np.uint8))
    with pytest.raises(RuntimeError):
        orb.extract(np.zeros((10, 10, 3), np.float64))
    with pytest.raises(RuntimeError):
        orb.extract(np.zeros((10, 10), np.uint8))        # BGR2GRAY on one channel fails in the reference too


def test_module_matcher_and_ratio_filter():
    import build.orb_project as orb
    a, b = _bgr(2), _bgr(2)
    b = np.roll(b, 1, axis=1)
    k1, d1 = orb.extract(a)
    k2, d2 = orb.extract(b)
    idx, dist = orb.knn_match2(d1, d2)
    eidx, edist = oracle.c.knn2(d1, d2)
    assert idx.dtype == np.int32 and np.array_equal(idx, eidx) and np.array_equal(dist, edist)
    gq, g2, g1 = orb.ratio_filter(idx, dist)
    e = oracle.c.ratio_filter(eidx, edist)
    assert np.array_equal(gq, e[0]) and np.array_equal(g2, e[1]) and np.array_equal(g1, e[2])
    # the (0, 0) descriptor array of an empty frame is accepted on either side
    empty = np.empty((0, 0), np.uint8)
    i0, d0 = orb.knn_match2(empty, d2)
    assert i0.shape == (0, 2)
    i1, dd1 = orb.knn_match2(d1[:3], empty)
    assert (i1 == -1).all() and (dd1 == -1).all()
    with pytest.raises(ValueError):
        orb.ratio_filter(i1, dd1)
    kp, desc, vidx = orb.extract_indexed(a)
    assert np.array_equal(desc, d1) and len(vidx) == len(desc)
    x, y = kp[vidx, 0], kp[vidx, 1]
    assert (x >= 16).all() and (y >= 16).all()


def test_host_mirror_of_reference_interface():
    from visual_slam_amd import orb
    img = _bgr(3)
    gray = oracle.c.bgr2gray(img)
    kp = orb.ORBDescriptor.detectKeypoints(gray)
    exy = oracle.c.detect(gray)
    assert np.array_equal(kp[:, :2].astype(np.int32), exy)
    assert (kp[:, 2] == 7).all() and (kp[:, 3] == -1).all() and (kp[:, 6] == -1).all()
    orb.ORBDescriptor.computeOrientation(gray, kp)
    assert np.array_equal(bits(kp[:, 3].copy()), bits(oracle.c.orient(gray, exy)))
    desc = orb.ORBDescriptor.computeDescriptors(gray, kp)
    ekp, edesc, _ = oracle.c.extract_gray(gray)
    assert np.array_equal(bits(kp), bits(ekp)) and np.array_equal(desc, edesc)
    kp2, desc2 = orb.extract(img)
    assert np.array_equal(bits(kp2), bits(ekp)) and np.array_equal(desc2, edesc)

    # the matcher as views.py uses it: knnMatch(d1, d2, 2) then the ratio loop (views.py:162-171)
    d_prev = oracle.c.extract_gray(np.roll(gray, 1, axis=1))[1]
    matcher = orb.BFMatcher(orb.NORM_HAMMING, crossCheck=False)
    matches = matcher.knnMatch(desc, d_prev, 2)
    eidx, edist = oracle.c.knn2(desc, d_prev)
    assert len(matches) == len(desc)
    for q, (m1, m2) in enumerate(matches):
        assert (m1.queryIdx, m1.trainIdx, m2.trainIdx) == (q, eidx[q, 0], eidx[q, 1])
        assert (m1.distance, m2.distance) == (float(edist[q, 0]), float(edist[q, 1])) and m1.imgIdx == 0
    f1, f2, first = orb.lowe_ratio_matches(matches)
    e = oracle.c.ratio_filter(eidx, edist)
    assert np.array_equal(f1, e[0]) and np.array_equal(f2, e[1]) and np.array_equal(first, e[2])
    g = orb.good_matches(desc, d_prev)
    assert all(np.array_equal(a, b) for a, b in zip(g, e))
    assert [m.trainIdx for m in matcher.match(desc, d_prev)] == eidx[:, 0].tolist()


def test_concurrent_calls_from_two_python_threads_serialise():
    """The reference's caller runs extract on a worker thread beside a GUI thread (main.py:85-86).
    Two threads calling into the module at once must take turns, not deadlock on the module's
    lock and the GIL."""
    import threading
    orb = importlib.import_module("build.orb_project")
    img = np.repeat(make_test_image(120, 160, 5)[:, :, None], 3, axis=2)
    kp0, d0 = orb.extract(img)
    assert len(d0) >= 2
    i0, s0 = orb.knn_match2(d0, d0)
    errors, done = [], []

    def extractor():
        try:
            for _ in range(40):
                kp, d = orb.extract(img)
                assert np.array_equal(d, d0) and np.array_equal(kp.view(np.uint32), kp0.view(np.uint32))
            done.append("e")
        except Exception as exc:      # pragma: no cover
            errors.append(exc)

    def matcher():
        try:
            for _ in range(40):
                i, s = orb.knn_match2(d0, d0)
                assert np.array_equal(i, i0) and np.array_equal(s, s0)
            done.append("m")
        except Exception as exc:      # pragma: no cover
            errors.append(exc)

    ts = [threading.Thread(target=extractor, daemon=True), threading.Thread(target=matcher, daemon=True)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    assert not errors and sorted(done) == ["e", "m"], "threads hung or failed: %r %r" % (done, errors)
