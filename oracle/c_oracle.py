"""ctypes binding of oracle/liborb_oracle.so (see orb_oracle.c).  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liborb_oracle.so")
_lib = None

_u8p = ctypes.POINTER(ctypes.c_uint8)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)


def build(force=False):
    """Compile the C restatement with gcc (seconds)."""
    src = os.path.join(_HERE, "orb_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liborb_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        L.orc_bgr2gray.argtypes = [_u8p, ctypes.c_int, ctypes.c_int, _u8p]
        L.orc_bgr2gray.restype = None
        L.orc_detect.argtypes = [_u8p, ctypes.c_int, ctypes.c_int, _i32p, ctypes.c_long]
        L.orc_detect.restype = ctypes.c_long
        L.orc_orient.argtypes = [_u8p, ctypes.c_int, ctypes.c_int, _i32p, ctypes.c_long, _f32p]
        L.orc_orient.restype = None
        L.orc_moments_i.argtypes = [_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _i64p]
        L.orc_moments_i.restype = None
        L.orc_angle_from_moments.argtypes = [ctypes.c_double] * 3
        L.orc_angle_from_moments.restype = ctypes.c_float
        L.orc_describe.argtypes = [_u8p, ctypes.c_int, ctypes.c_int, _i32p, _f32p, ctypes.c_long, _u8p, _i32p]
        L.orc_describe.restype = ctypes.c_long
        L.orc_extract.argtypes = [_u8p, ctypes.c_int, ctypes.c_int, _f32p, ctypes.c_long, _u8p, _i32p,
                                  ctypes.POINTER(ctypes.c_long)]
        L.orc_extract.restype = ctypes.c_long
        L.orc_knn2.argtypes = [_u8p, ctypes.c_long, _u8p, ctypes.c_long, _i32p, _i32p]
        L.orc_knn2.restype = None
        L.orc_ratio_filter.argtypes = [_i32p, _i32p, ctypes.c_long, _i32p, _i32p, _i32p]
        L.orc_ratio_filter.restype = ctypes.c_long
        L.orc_pattern_points.argtypes = []
        L.orc_pattern_points.restype = ctypes.POINTER(ctypes.c_int8)
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def _gray(g):
    g = np.ascontiguousarray(g, dtype=np.uint8)
    assert g.ndim == 2
    return g


def bgr2gray(bgr):
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    H, W, C = bgr.shape
    assert C == 3
    out = np.empty((H, W), np.uint8)
    lib().orc_bgr2gray(_p(bgr, _u8p), H, W, _p(out, _u8p))
    return out


def detect(gray):
    """-> xy int32 [N,2] (x, y) in scan order."""
    g = _gray(gray)
    H, W = g.shape
    cap = 4096
    while True:
        xy = np.empty((cap, 2), np.int32)
        n = lib().orc_detect(_p(g, _u8p), H, W, _p(xy, _i32p), cap)
        if n <= cap:
            return xy[:n].copy()
        cap = int(n)


def moments(gray, x, y):
    g = _gray(gray)
    out = np.zeros(3, np.int64)
    lib().orc_moments_i(_p(g, _u8p), g.shape[0], g.shape[1], int(x), int(y), _p(out, _i64p))
    return out


def angle_from_moments(m00, m10, m01):
    return np.float32(lib().orc_angle_from_moments(float(m00), float(m10), float(m01)))


def orient(gray, xy):
    g = _gray(gray)
    xy = np.ascontiguousarray(xy, np.int32).reshape(-1, 2)
    ang = np.empty(len(xy), np.float32)
    lib().orc_orient(_p(g, _u8p), g.shape[0], g.shape[1], _p(xy, _i32p), len(xy), _p(ang, _f32p))
    return ang


def describe(gray, xy, angle):
    """-> (desc uint8 [Nv,32], valid_idx int32 [Nv])."""
    g = _gray(gray)
    xy = np.ascontiguousarray(xy, np.int32).reshape(-1, 2)
    angle = np.ascontiguousarray(angle, np.float32)
    n = len(xy)
    desc = np.empty((max(n, 1), 32), np.uint8)
    vidx = np.empty(max(n, 1), np.int32)
    nv = lib().orc_describe(_p(g, _u8p), g.shape[0], g.shape[1], _p(xy, _i32p), _p(angle, _f32p), n,
                            _p(desc, _u8p), _p(vidx, _i32p))
    return desc[:nv].copy(), vidx[:nv].copy()


def extract_gray(gray):
    """-> (kp float32 [N,7], desc uint8 [Nv,32], valid_idx int32 [Nv]); gray-frame part of extract()."""
    g = _gray(gray)
    H, W = g.shape
    cap = 8192
    while True:
        kp = np.empty((cap, 7), np.float32)
        desc = np.empty((cap, 32), np.uint8)
        vidx = np.empty(cap, np.int32)
        nv = ctypes.c_long(0)
        n = lib().orc_extract(_p(g, _u8p), H, W, _p(kp, _f32p), cap, _p(desc, _u8p), _p(vidx, _i32p),
                              ctypes.byref(nv))
        if n <= cap:
            return kp[:n].copy(), desc[:nv.value].copy(), vidx[:nv.value].copy()
        cap = int(n)


def extract_gray_unoptimised_seconds(frames):
    """Wall time of the gray-frame part of extract() over `frames` with the -O0 build of the same
    C file (the reference compiles its src/orb.cpp without optimisation, CMakeLists.txt:13).  For
    the CPU baseline only; the results are not returned."""
    import time
    so = os.path.join(_HERE, "liborb_oracle_O0.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liborb_oracle_O0.so"])
    L = ctypes.CDLL(so)
    L.orc_extract.argtypes = [_u8p, ctypes.c_int, ctypes.c_int, _f32p, ctypes.c_long, _u8p, _i32p,
                              ctypes.POINTER(ctypes.c_long)]
    L.orc_extract.restype = ctypes.c_long
    cap = 65536
    kp = np.empty((cap, 7), np.float32)
    desc = np.empty((cap, 32), np.uint8)
    vidx = np.empty(cap, np.int32)
    nv = ctypes.c_long(0)
    t0 = time.perf_counter()
    for g in frames:
        g = _gray(g)
        L.orc_extract(_p(g, _u8p), g.shape[0], g.shape[1], _p(kp, _f32p), cap, _p(desc, _u8p), _p(vidx, _i32p),
                      ctypes.byref(nv))
    return time.perf_counter() - t0


def extract(image_bgr):
    """The reference's extract(image) for [H,W,3] uint8 input (src/main_interface.cpp:80-101)."""
    kp, desc, _ = extract_gray(bgr2gray(image_bgr))
    if len(kp) == 0:
        desc = np.empty((0, 0), np.uint8)
    return kp, desc


def knn2(q, t):
    """-> (idx int32 [nq,2], dist int32 [nq,2]); -1 where fewer than two train rows exist."""
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    idx = np.empty((len(q), 2), np.int32)
    dist = np.empty((len(q), 2), np.int32)
    lib().orc_knn2(_p(q, _u8p), len(q), _p(t, _u8p), len(t), _p(idx, _i32p), _p(dist, _i32p))
    return idx, dist


def ratio_filter(idx, dist):
    """-> (good_q, good_t2, good_t1) as views.py:167-171 builds them; raises like the reference
    when a query has fewer than two neighbours."""
    idx = np.ascontiguousarray(idx, np.int32)
    dist = np.ascontiguousarray(dist, np.int32)
    nq = len(idx)
    gq = np.empty(max(nq, 1), np.int32)
    g2 = np.empty(max(nq, 1), np.int32)
    g1 = np.empty(max(nq, 1), np.int32)
    m = lib().orc_ratio_filter(_p(idx, _i32p), _p(dist, _i32p), nq, _p(gq, _i32p), _p(g2, _i32p), _p(g1, _i32p))
    if m < 0:
        raise ValueError("not enough values to unpack (expected 2)")
    return gq[:m].copy(), g2[:m].copy(), g1[:m].copy()


def pattern_points():
    """The restatement's own table: 512 points (x, y) in the reference's order."""
    p = lib().orc_pattern_points()
    return np.ctypeslib.as_array(p, shape=(512, 2)).copy()


def pattern_pairs():
    """The same as 256 test pairs (x1, y1, x2, y2): pair k = point k against point k + 256."""
    pts = pattern_points()
    return np.concatenate([pts[:256], pts[256:]], axis=1)
