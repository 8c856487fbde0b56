"""Detector extension (arc length, wrap-around, threshold, score + 3x3 non-maximum suppression): an
opt-in the reference does not have.  Its definition is oracle/np_ref.py detect_ext; with the
reference's parameters it must be the reference's detector.  GPU tests compare the library with it."""
import numpy as np
import pytest

import oracle
from conftest import make_test_image
from oracle import np_ref

COMBOS = [dict(threshold=20, arc=9, wrap=True, nms=False), dict(threshold=20, arc=9, wrap=True, nms=True),
          dict(threshold=20, arc=12, wrap=True, nms=False), dict(threshold=35, arc=12, wrap=False, nms=False),
          dict(threshold=20, arc=12, wrap=False, nms=True), dict(threshold=10, arc=10, wrap=False, nms=True),
          dict(threshold=25, arc=16, wrap=True, nms=False)]


def test_extension_with_reference_parameters_is_the_reference_detector():
    for seed, (H, W) in enumerate([(90, 120), (61, 83), (240, 320)]):
        g = make_test_image(H, W, 40 + seed)
        xy, score = np_ref.detect_ext(g)
        assert np.array_equal(xy, oracle.c.detect(g)) and np.array_equal(xy, np_ref.detect(g))
        assert (score >= 20).all()


def test_extension_semantics_on_hand_made_cases():
    g = np.full((40, 40), 100, np.uint8)
    g[20, 20] = 200                                   # all 16 neighbours darker by 100
    xy, score = np_ref.detect_ext(g, arc=9, wrap=True, nms=True)
    assert xy.tolist() == [[20, 20]] and score.tolist() == [100]
    # a run k = 10 .. 15, 0 .. 5 (12 long) only exists with wrap-around
    g = np.full((40, 40), 100, np.uint8)
    for k in list(range(10, 16)) + list(range(0, 6)):
        dx, dy = np_ref._CIRCLE[k]
        g[20 + dy, 20 + dx] = 150
    assert [20, 20] not in np_ref.detect_ext(g, arc=12, wrap=False)[0].tolist()
    assert [20, 20] in np_ref.detect_ext(g, arc=12, wrap=True)[0].tolist()
    # a plateau of equal scores is suppressed entirely (strict comparison)
    g = np.full((40, 40), 100, np.uint8)
    g[20, 20:22] = 200
    kept = np_ref.detect_ext(g, arc=9, wrap=True, nms=True)[0].tolist()
    assert [20, 20] not in kept and [21, 20] not in kept


@pytest.mark.gpu
@pytest.mark.parametrize("combo", COMBOS)
def test_gpu_detector_extension_equals_its_definition(gctx, combo):
    try:
        gctx.set_detector(**combo)
        for seed, (H, W) in enumerate([(150, 210), (61, 83), (480, 640), (7, 7), (64, 257)]):
            g = make_test_image(H, W, 70 + seed)
            want, _ = np_ref.detect_ext(g, **combo)
            assert np.array_equal(gctx.detect(g), want), (combo, H, W)
    finally:
        gctx.set_detector()
    g = make_test_image(100, 130, 3)
    assert np.array_equal(gctx.detect(g), oracle.c.detect(g))      # back on the reference's detector


@pytest.mark.gpu
def test_gpu_extract_with_extension_detector_orients_and_describes_as_always(gctx):
    g = make_test_image(200, 260, 9)
    combo = dict(threshold=20, arc=9, wrap=True, nms=True)
    try:
        gctx.set_detector(**combo)
        kp, desc, vidx = gctx.extract_gray(g)
    finally:
        gctx.set_detector()
    xy, _ = np_ref.detect_ext(g, **combo)
    assert len(xy) > 50 and np.array_equal(kp[:, :2].astype(np.int32), xy)
    ang = oracle.c.orient(g, xy)
    assert np.array_equal(kp[:, 3].view(np.uint32), ang.view(np.uint32))
    edesc, evidx = oracle.c.describe(g, xy, ang)
    assert np.array_equal(desc, edesc) and np.array_equal(vidx, evidx)
    with pytest.raises(Exception):
        gctx.set_detector(arc=8)
