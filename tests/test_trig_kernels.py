"""The purpose-built arc tangent / sine / cosine of the orientation step (visual_slam_amd/csrc/orb_trig.h), run on
the host (no GPU): the header is plain C++ with __builtin_fma, tests/trig_host.cpp wraps it, and g++ -mfma gives the
arithmetic the device executes.  What the kernels rely on:

* atan2_deg_moments: relative error of a few units in the last place, far inside the 2^-43 window its `unsure`
  flag reports, so that the narrowing to float (src/orb.cpp:129) gives the float any libm within an ulp gives;
* sincos_turn: below one unit in the last place on [0, 2 pi], i.e. as close to the reference's libm as another
  libm would be (src/orb.cpp:173-174).
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PI = 3.1415926535897932384626433832795


def _has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read()
    except OSError:
        return False


@pytest.fixture(scope="module")
def trig():
    if not _has_fma():
        pytest.skip("host CPU has no FMA instruction")
    so = os.path.join(ROOT, "build", "libtrig_host.so")
    src = os.path.join(ROOT, "tests", "trig_host.cpp")
    hdr = os.path.join(ROOT, "visual_slam_amd", "csrc", "orb_trig.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-mfma", "-ffp-contract=off", "-shared", "-fPIC", "-o", so, src, "-lm"])
    return ctypes.CDLL(so)


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def _moments(n, seed):
    rng = np.random.default_rng(seed)
    lim = 31 * 31 * 255 * 15 // 4          # well beyond what a 31 x 31 patch of bytes can produce per axis weight
    m01 = rng.integers(-lim, lim, n).astype(np.int32)
    m10 = rng.integers(-lim, lim, n).astype(np.int32)
    k = n // 10
    m01[:2 * k] = rng.integers(-300, 300, 2 * k)          # small moments, one or both
    m10[:k] = rng.integers(-300, 300, k)
    m01[2 * k:3 * k] = 0                                  # the axes
    m10[3 * k:4 * k] = 0
    m01[4 * k:5 * k] = m10[4 * k:5 * k]                   # the diagonals
    m01[5 * k:6 * k] = -m10[5 * k:6 * k]
    return m01, m10


def test_atan2_series_accuracy_and_narrowing(trig):
    n = 1_000_000
    m01, m10 = _moments(n, 11)
    deg = np.empty(n)
    unsure = np.empty(n, np.uint8)
    trig.trig_atan2_deg(_p(m01, ctypes.c_int32), _p(m10, ctypes.c_int32), ctypes.c_long(n), _p(deg, ctypes.c_double),
                        _p(unsure, ctypes.c_uint8))
    # 80-bit reference of the same formula: atan2 times the double constant 180 / CV_PI
    ref = np.arctan2(m01.astype(np.longdouble), m10.astype(np.longdouble)) * np.longdouble(180.0 / PI)
    nz = ref != 0
    assert np.array_equal(deg[~nz], np.zeros((~nz).sum()))
    rel = np.abs((deg.astype(np.longdouble) - ref)[nz] / ref[nz])
    assert float(rel.max()) < 2.0 ** -50, float(rel.max())
    assert unsure.sum() < 50                                # about one in 260 000 expected
    # the stored float angle equals the one this host's libm gives wherever the flag is clear (where it is set, the
    # two may differ: either could be the reference's float, that is what the flag means)
    mine = np.empty(n, np.uint32)
    libm = np.empty(n, np.uint32)
    trig.trig_angle_floats(_p(m01, ctypes.c_int32), _p(m10, ctypes.c_int32), ctypes.c_long(n), _p(mine, ctypes.c_uint32),
                           _p(libm, ctypes.c_uint32), _p(unsure, ctypes.c_uint8))
    clear = unsure == 0
    assert np.array_equal(mine[clear], libm[clear])


def test_atan2_series_special_cases(trig):
    cases = [(0, 0, 0.0), (0, 5, 0.0), (0, -5, 180.0), (7, 0, 90.0), (-7, 0, -90.0), (3, 3, 45.0), (3, -3, 135.0),
             (-3, -3, -135.0), (-3, 3, -45.0), (1, 4000000, None), (4000000, 1, None), (-1, -4000000, None)]
    m01 = np.array([c[0] for c in cases], np.int32)
    m10 = np.array([c[1] for c in cases], np.int32)
    deg = np.empty(len(cases))
    unsure = np.empty(len(cases), np.uint8)
    trig.trig_atan2_deg(_p(m01, ctypes.c_int32), _p(m10, ctypes.c_int32), ctypes.c_long(len(cases)),
                        _p(deg, ctypes.c_double), _p(unsure, ctypes.c_uint8))
    for (y, x, want), got in zip(cases, deg):
        ref = float(np.arctan2(np.longdouble(y), np.longdouble(x)) * np.longdouble(180.0 / PI))
        assert np.float32(got) == np.float32(ref), (y, x, got, ref)
        if want is not None:
            assert np.float32(got) == np.float32(want), (y, x, got)


def test_sincos_below_one_ulp(trig):
    mpmath = pytest.importorskip("mpmath")
    mpmath.mp.prec = 200
    rng = np.random.default_rng(5)
    ang = np.concatenate([rng.uniform(0, 360, 4000), np.arange(0, 361, 15, dtype=np.float64),
                          np.nextafter(np.float32([90, 180, 270, 360]), np.float32(0)).astype(np.float64),
                          [1e-30, 1e-6, 44.999996, 45.000004, 359.99997]]).astype(np.float32)
    x = ang.astype(np.float64) * (PI / 180.0)
    s = np.empty_like(x)
    c = np.empty_like(x)
    trig.trig_sincos(_p(x, ctypes.c_double), ctypes.c_long(x.size), _p(s, ctypes.c_double), _p(c, ctypes.c_double))
    worst = 0.0
    for i in range(x.size):
        xi = mpmath.mpf(float(x[i]))
        for got, true in ((s[i], mpmath.sin(xi)), (c[i], mpmath.cos(xi))):
            # absolute error in units of the last place of a result of magnitude >= 2^-53 (what the rotation
            # products can feel: they are added to terms of magnitude up to 15)
            ulp = max(float(np.spacing(abs(float(true)))), 2.0 ** -105)
            worst = max(worst, float(abs(mpmath.mpf(float(got)) - true) / ulp))
    assert worst < 1.0, worst


def test_sincos_against_host_libm_on_a_dense_range(trig):
    """4 million consecutive float angles: the series and this host's libm differ by at most one unit in the
    last place, and only for a few per cent of the angles."""
    out = (ctypes.c_long * 4)()
    lo = int(np.float32(123.0).view(np.uint32))
    trig.trig_sincos_vs_libm(ctypes.c_uint32(lo), ctypes.c_uint32(lo + 4_000_000), out)
    n, ds, dc, worst = list(out)
    assert n == 4_000_000 and worst <= 1
    assert ds < 0.05 * n and dc < 0.05 * n


@pytest.mark.gpu
def test_device_trigonometry_is_the_host_build_bit_for_bit(trig, gctx):
    """The same header on the device (through the library's test hook) and on the host: identical doubles, so what
    the host-side tests and scripts/trig_exhaustive.py establish holds for the kernels."""
    import torch
    n = 2_000_000
    m01, m10 = _moments(n, 23)
    rng = np.random.default_rng(29)
    ang = np.concatenate([rng.uniform(0, 360, n - 400), np.arange(0, 361, dtype=np.float64),
                          np.nextafter(np.float32([90, 180, 270, 360]), np.float32(0)).astype(np.float64),
                          rng.uniform(0, 1e-3, 35)]).astype(np.float32)
    rad = ang.astype(np.float64) * (PI / 180.0)
    m = rad.size
    deg_h = np.empty(n)
    uns_h = np.empty(n, np.uint8)
    trig.trig_atan2_deg(_p(m01, ctypes.c_int32), _p(m10, ctypes.c_int32), ctypes.c_long(n), _p(deg_h, ctypes.c_double),
                        _p(uns_h, ctypes.c_uint8))
    s_h = np.empty(m)
    c_h = np.empty(m)
    trig.trig_sincos(_p(rad, ctypes.c_double), ctypes.c_long(m), _p(s_h, ctypes.c_double), _p(c_h, ctypes.c_double))
    dev = torch.device("cuda:0")
    d01 = torch.from_numpy(m01).to(dev)
    d10 = torch.from_numpy(m10).to(dev)
    drad = torch.from_numpy(rad).to(dev)
    ddeg = torch.empty(n, dtype=torch.float64, device=dev)
    duns = torch.empty(n, dtype=torch.uint8, device=dev)
    dsin = torch.empty(m, dtype=torch.float64, device=dev)
    dcos = torch.empty(m, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    gctx.trig_selftest_dev(d01.data_ptr(), d10.data_ptr(), n, ddeg.data_ptr(), duns.data_ptr(), drad.data_ptr(), m,
                           dsin.data_ptr(), dcos.data_ptr())
    gctx.sync()
    assert np.array_equal(ddeg.cpu().numpy().view(np.uint64), deg_h.view(np.uint64))
    assert np.array_equal(duns.cpu().numpy(), uns_h)
    assert np.array_equal(dsin.cpu().numpy().view(np.uint64), s_h.view(np.uint64))
    assert np.array_equal(dcos.cpu().numpy().view(np.uint64), c_h.view(np.uint64))
