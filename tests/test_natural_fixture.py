"""CPU: the natural-texture fixture (tests/golden/natural_pair.npz, made by
tests/golden/make_natural_fixture.py) against the CPU restatement."""
import hashlib
import os

import numpy as np

import oracle

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "natural_pair.npz")


def sha(*arrays):
    return np.frombuffer(hashlib.sha256(b"".join(np.ascontiguousarray(a).tobytes() for a in arrays)).digest(), np.uint8)


def check_natural_pair(extract, knn2, ratio_filter):
    """extract(bgr) -> (kp, desc); knn2(q, t) -> (idx, dist); ratio_filter(idx, dist) -> (good_q, ...)."""
    z = np.load(FIX)
    desc = {}
    for name in ("a", "b"):
        kp, d = extract(z["bgr_" + name])
        assert (len(kp), len(d)) == (int(z["n_" + name]), int(z["nv_" + name]))
        assert np.array_equal(sha(kp), z["kp_sha_" + name]) and np.array_equal(sha(d), z["desc_sha_" + name])
        desc[name] = d
    idx, dist = knn2(desc["b"], desc["a"])
    assert np.array_equal(sha(idx, dist), z["match_sha"])
    assert np.array_equal(ratio_filter(idx, dist)[0], z["good"])


def test_oracle_reproduces_natural_fixture():
    check_natural_pair(oracle.c.extract, oracle.c.knn2, oracle.c.ratio_filter)
