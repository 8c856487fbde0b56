"""ctypes binding of liborbamd.so (C ABI declared in include/orbamd.h).

The library is the only compute back end of this package.  If it is missing or no GPU is usable
the failure is an exception, never a CPU substitute.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ORBAMD_LIB: another build of the same library (kernel experiments); default is the in-tree build
LIB_PATH = os.environ.get("ORBAMD_LIB") or os.path.join(_HERE, "liborbamd.so")

OK, E_BADARG, E_CAPACITY, E_DEVICE, E_NOMEM, E_STATE = 0, -1, -2, -3, -4, -5
MAX_TRAIN = 1 << 18
NOALLOC_MAX_QUERIES = 65536
K_COUNT = 10
MATCHER_MATRIX, MATCHER_VALU = 0, 1

_c_ctx = ctypes.c_void_p
_u8p = ctypes.POINTER(ctypes.c_uint8)
_i32p = ctypes.POINTER(ctypes.c_int32)
_f32p = ctypes.POINTER(ctypes.c_float)
_intp = ctypes.POINTER(ctypes.c_int)
_vp = ctypes.c_void_p
_int = ctypes.c_int

# name -> (restype, argtypes); every symbol include/orbamd.h declares
SIGNATURES = {
    "orbamd_version": (_int, []),
    "orbamd_status_string": (ctypes.c_char_p, [_int]),
    "orbamd_ctx_create": (_int, [_int, ctypes.POINTER(_c_ctx)]),
    "orbamd_ctx_destroy": (_int, [_c_ctx]),
    "orbamd_ctx_set_stream": (_int, [_c_ctx, _vp]),
    "orbamd_ctx_use_own_stream": (_int, [_c_ctx]),
    "orbamd_ctx_reserve": (_int, [_c_ctx, _int, _int, _int, _int]),
    "orbamd_ctx_sync": (_int, [_c_ctx]),
    "orbamd_last_error": (ctypes.c_char_p, [_c_ctx]),
    "orbamd_ctx_set_matcher": (_int, [_c_ctx, _int]),
    "orbamd_ctx_set_detector": (_int, [_c_ctx, _int, _int, _int, _int]),
    "orbamd_ctx_set_audit": (_int, [_c_ctx, _int]),
    "orbamd_ctx_audit_read": (_int, [_c_ctx, _vp, _int]),
    "orbamd_ctx_timing_enable": (_int, [_c_ctx, _int]),
    "orbamd_ctx_timing_reset": (_int, [_c_ctx]),
    "orbamd_ctx_timing_read": (_int, [_c_ctx, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long)]),
    "orbamd_kernel_name": (ctypes.c_char_p, [_int]),
    "orbamd_ctx_host_timing_read": (_int, [_c_ctx, ctypes.POINTER(ctypes.c_double)]),
    "orbamd_bgr2gray": (_int, [_c_ctx, _vp, _int, _int, _vp]),
    "orbamd_detect": (_int, [_c_ctx, _vp, _int, _int, _vp, _int, _intp]),
    "orbamd_orient": (_int, [_c_ctx, _vp, _int, _int, _vp, _int, _vp]),
    "orbamd_describe": (_int, [_c_ctx, _vp, _int, _int, _vp, _vp, _int, _vp, _vp, _intp]),
    "orbamd_extract_gray": (_int, [_c_ctx, _vp, _int, _int, _intp, _intp]),
    "orbamd_extract_bgr": (_int, [_c_ctx, _vp, _int, _int, _intp, _intp]),
    "orbamd_fetch": (_int, [_c_ctx, _vp, _vp, _vp]),
    "orbamd_knn2": (_int, [_c_ctx, _vp, _int, _vp, _int, _vp, _vp]),
    "orbamd_ratio_filter": (_int, [_c_ctx, _vp, _vp, _int, _vp, _vp, _vp, _intp]),
    "orbamd_knn2_ratio": (_int, [_c_ctx, _vp, _int, _vp, _int, _vp, _vp, _vp, _vp, _vp, _intp]),
    "orbamd_synth_frames_dev": (_int, [_c_ctx, _vp, _int, _int, _int, _int, ctypes.c_uint32, _int, _int]),
    "orbamd_synth_textured_dev": (_int, [_c_ctx, _vp, _int, _int, _int, _int, ctypes.c_uint32]),
    "orbamd_bgr2gray_dev": (_int, [_c_ctx, _vp, _int, _int, _int, _vp]),
    "orbamd_extract_batch_dev": (_int, [_c_ctx, _vp, _int, _int, _int, _int, _vp, _vp, _vp, _vp]),
    "orbamd_knn2_batch_dev": (_int, [_c_ctx, _vp, _vp, _int, _int, _vp, _int, _vp, _vp, _vp]),
    "orbamd_ratio_filter_batch_dev": (_int, [_c_ctx, _vp, _vp, _vp, _int, _int, _vp, _int, _vp, _vp, _vp, _vp]),
    "orbamd_knn2_cross_dev": (_int, [_c_ctx, _vp, _vp, _int, _vp, _vp, _int, _int, _vp, _vp, _vp]),
    "orbamd_knn2_dev": (_int, [_c_ctx, _vp, _int, _vp, _int, _vp, _vp, _vp]),
    "orbamd_knn2_dev_noalloc": (_int, [_c_ctx, _vp, _int, _vp, _int, _vp, _vp, _vp]),
    "orbamd_trig_selftest_dev": (_int, [_c_ctx, _vp, _vp, _int, _vp, _vp, _vp, _int, _vp, _vp]),
}

_lib = None


class OrbAmdError(RuntimeError):
    def __init__(self, status, what, detail=""):
        self.status = status
        msg = "%s: %s" % (what, status_string(status))
        if detail:
            msg += " [%s]" % detail
        super().__init__(msg)


def _prefer_torch_hip_runtime():
    """liborbamd.so needs libamdhip64.so.7.  A process can hold only one HIP runtime: if PyTorch is
    installed and is imported AFTER this library has pulled in the system runtime, torch finds a
    runtime it was not built with and reports no GPU.  So when torch is installed (imported or not),
    load ITS bundled runtime first; the dynamic loader then resolves our dependency to that copy.
    Without torch the system runtime under /opt/rocm is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load liborbamd.so (built in-tree by `make`); raises if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "visual_slam_amd: %s not found -- build it with `make` at the repository root "
                "(hipcc, gfx950).  There is no CPU fallback." % LIB_PATH)
        _prefer_torch_hip_runtime()
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here = header and library disagree
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def status_string(status):
    return lib().orbamd_status_string(int(status)).decode()


def _np_ptr(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else None


class Context:
    """One orbamd context: a HIP stream plus device work buffers on one GPU."""

    def __init__(self, device=0):
        self._h = _c_ctx()
        st = lib().orbamd_ctx_create(int(device), ctypes.byref(self._h))
        if st != OK:
            self._h = None
            raise OrbAmdError(st, "orbamd_ctx_create(device=%d) -- a usable AMD GPU is required" % device)
        self.device = int(device)

    # -- plumbing -----------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            lib().orbamd_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, st, what, allow=()):
        if st != OK and st not in allow:
            detail = lib().orbamd_last_error(self._h).decode() if st == E_DEVICE else ""
            raise OrbAmdError(st, what, detail)
        return st

    def set_stream(self, stream_ptr):
        """Run all later work on this hipStream_t handle; 0 / None is HIP's default (null) stream."""
        self._chk(lib().orbamd_ctx_set_stream(self._h, ctypes.c_void_p(stream_ptr or 0)), "set_stream")

    def use_own_stream(self):
        self._chk(lib().orbamd_ctx_use_own_stream(self._h), "use_own_stream")

    def reserve(self, frames, H, W, cap):
        """Allocate the work buffers for batches of this size now, so that later calls only enqueue."""
        self._chk(lib().orbamd_ctx_reserve(self._h, int(frames), int(H), int(W), int(cap)), "reserve")

    def sync(self):
        self._chk(lib().orbamd_ctx_sync(self._h), "sync")

    def set_matcher(self, matcher):
        """MATCHER_MATRIX (default) or MATCHER_VALU: the kernel family used by every knn2 call."""
        self._chk(lib().orbamd_ctx_set_matcher(self._h, int(matcher)), "set_matcher")

    def set_detector(self, threshold=20, arc=12, wrap=False, nms=False):
        """Detector extension (off by default = the reference's detector): see include/orbamd.h."""
        self._chk(lib().orbamd_ctx_set_detector(self._h, int(threshold), int(arc), 1 if wrap else 0,
                                                1 if nms else 0), "set_detector")

    def set_audit(self, on=True):
        """Parity audit (see include/orbamd.h): count narrowings that a different libm could round the other way."""
        self._chk(lib().orbamd_ctx_set_audit(self._h, 1 if on else 0), "set_audit")

    def audit_read(self, frames=1):
        """-> int32 [frames, 4]: (angles, coordinates near a float midpoint, decisive coordinates, 0) per frame of
        the last orient / describe / extract call."""
        out = np.zeros((frames, 4), np.int32)
        self._chk(lib().orbamd_ctx_audit_read(self._h, _np_ptr(out), int(frames)), "audit_read")
        return out

    def timing_enable(self, on=True):
        self._chk(lib().orbamd_ctx_timing_enable(self._h, 1 if on else 0), "timing_enable")

    def timing_reset(self):
        self._chk(lib().orbamd_ctx_timing_reset(self._h), "timing_reset")

    def timing_read(self):
        """-> {kernel name: (milliseconds, launches)} since the last reset."""
        ms = (ctypes.c_double * K_COUNT)()
        n = (ctypes.c_long * K_COUNT)()
        self._chk(lib().orbamd_ctx_timing_read(self._h, ms, n), "timing_read")
        return {lib().orbamd_kernel_name(k).decode(): (ms[k], n[k]) for k in range(K_COUNT)}

    def host_timing_read(self):
        """-> {"h2d", "enqueue", "wait", "d2h"}: host microseconds inside the single-frame calls since the reset."""
        us = (ctypes.c_double * 4)()
        self._chk(lib().orbamd_ctx_host_timing_read(self._h, us), "host_timing_read")
        return dict(zip(("h2d", "enqueue", "wait", "d2h"), (float(v) for v in us)))

    # -- single frame, host arrays ----------------------------------------------------------------
    @staticmethod
    def _gray(g):
        g = np.ascontiguousarray(g)
        if g.dtype != np.uint8 or g.ndim != 2:
            raise ValueError("gray image must be uint8 [H, W]")
        return g

    def bgr2gray(self, bgr):
        bgr = np.ascontiguousarray(bgr)
        if bgr.dtype != np.uint8 or bgr.ndim != 3 or bgr.shape[2] != 3:
            raise ValueError("BGR image must be uint8 [H, W, 3]")
        H, W = bgr.shape[:2]
        out = np.empty((H, W), np.uint8)
        self._chk(lib().orbamd_bgr2gray(self._h, _np_ptr(bgr), H, W, _np_ptr(out)), "bgr2gray")
        return out

    def detect(self, gray, cap=None):
        """-> xy int32 [N, 2]; with cap given, raises OrbAmdError(E_CAPACITY) if N > cap."""
        g = self._gray(gray)
        H, W = g.shape
        n = ctypes.c_int(0)
        if cap is not None:
            xy = np.empty((max(cap, 1), 2), np.int32)
            self._chk(lib().orbamd_detect(self._h, _np_ptr(g), H, W, _np_ptr(xy), int(cap), ctypes.byref(n)),
                      "detect")
            return xy[:n.value].copy()
        guess = 16384
        while True:
            xy = np.empty((guess, 2), np.int32)
            st = self._chk(lib().orbamd_detect(self._h, _np_ptr(g), H, W, _np_ptr(xy), guess, ctypes.byref(n)),
                           "detect", allow=(E_CAPACITY,))
            if st == OK:
                return xy[:n.value].copy()
            guess = n.value

    def orient(self, gray, xy):
        g = self._gray(gray)
        xy = np.ascontiguousarray(xy, np.int32).reshape(-1, 2)
        ang = np.empty(len(xy), np.float32)
        self._chk(lib().orbamd_orient(self._h, _np_ptr(g), g.shape[0], g.shape[1], _np_ptr(xy), len(xy),
                                      _np_ptr(ang)), "orient")
        return ang

    def describe(self, gray, xy, angle):
        g = self._gray(gray)
        xy = np.ascontiguousarray(xy, np.int32).reshape(-1, 2)
        angle = np.ascontiguousarray(angle, np.float32).reshape(-1)
        if len(angle) != len(xy):
            raise ValueError("one angle per keypoint")
        n = len(xy)
        desc = np.empty((max(n, 1), 32), np.uint8)
        vidx = np.empty(max(n, 1), np.int32)
        nv = ctypes.c_int(0)
        self._chk(lib().orbamd_describe(self._h, _np_ptr(g), g.shape[0], g.shape[1], _np_ptr(xy), _np_ptr(angle),
                                        n, _np_ptr(desc), _np_ptr(vidx), ctypes.byref(nv)), "describe")
        return desc[:nv.value].copy(), vidx[:nv.value].copy()

    def _fetch(self, n, nv):
        kp = np.empty((n, 7), np.float32)
        desc = np.empty((nv, 32), np.uint8)
        vidx = np.empty(nv, np.int32)
        self._chk(lib().orbamd_fetch(self._h, _np_ptr(kp), _np_ptr(desc), _np_ptr(vidx)), "fetch")
        return kp, desc, vidx

    def extract_gray(self, gray):
        """-> (kp float32 [N,7], desc uint8 [Nv,32], valid_idx int32 [Nv])."""
        g = self._gray(gray)
        n, nv = ctypes.c_int(0), ctypes.c_int(0)
        self._chk(lib().orbamd_extract_gray(self._h, _np_ptr(g), g.shape[0], g.shape[1], ctypes.byref(n),
                                            ctypes.byref(nv)), "extract_gray")
        return self._fetch(n.value, nv.value)

    def extract_bgr(self, bgr):
        bgr = np.ascontiguousarray(bgr)
        if bgr.dtype != np.uint8 or bgr.ndim != 3 or bgr.shape[2] != 3:
            raise ValueError("BGR image must be uint8 [H, W, 3]")
        n, nv = ctypes.c_int(0), ctypes.c_int(0)
        self._chk(lib().orbamd_extract_bgr(self._h, _np_ptr(bgr), bgr.shape[0], bgr.shape[1], ctypes.byref(n),
                                           ctypes.byref(nv)), "extract_bgr")
        return self._fetch(n.value, nv.value)

    def knn2(self, q, t):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        idx = np.empty((len(q), 2), np.int32)
        dist = np.empty((len(q), 2), np.int32)
        self._chk(lib().orbamd_knn2(self._h, _np_ptr(q), len(q), _np_ptr(t), len(t), _np_ptr(idx), _np_ptr(dist)),
                  "knn2")
        return idx, dist

    def knn2_ratio(self, q, t):
        """knn2 + ratio_filter in one call: (idx, dist, good_q, good_t2, good_t1)."""
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        nq = len(q)
        idx = np.empty((nq, 2), np.int32)
        dist = np.empty((nq, 2), np.int32)
        gq, g2, g1 = (np.empty(max(nq, 1), np.int32) for _ in range(3))
        m = ctypes.c_int(0)
        st = self._chk(lib().orbamd_knn2_ratio(self._h, _np_ptr(q), nq, _np_ptr(t), len(t), _np_ptr(idx), _np_ptr(dist),
                                               _np_ptr(gq), _np_ptr(g2), _np_ptr(g1), ctypes.byref(m)),
                       "knn2_ratio", allow=(E_BADARG,))
        if st == E_BADARG:
            if m.value == -1:
                raise ValueError("not enough values to unpack (expected 2)")
            raise OrbAmdError(st, "knn2_ratio")
        return idx, dist, gq[:m.value].copy(), g2[:m.value].copy(), g1[:m.value].copy()

    def ratio_filter(self, idx, dist):
        idx = np.ascontiguousarray(idx, np.int32).reshape(-1, 2)
        dist = np.ascontiguousarray(dist, np.int32).reshape(-1, 2)
        nq = len(idx)
        gq = np.empty(max(nq, 1), np.int32)
        g2 = np.empty(max(nq, 1), np.int32)
        g1 = np.empty(max(nq, 1), np.int32)
        m = ctypes.c_int(0)
        st = self._chk(lib().orbamd_ratio_filter(self._h, _np_ptr(idx), _np_ptr(dist), nq, _np_ptr(gq), _np_ptr(g2),
                                                 _np_ptr(g1), ctypes.byref(m)), "ratio_filter", allow=(E_BADARG,))
        if st == E_BADARG:
            raise ValueError("not enough values to unpack (expected 2)")
        return gq[:m.value].copy(), g2[:m.value].copy(), g1[:m.value].copy()

    # -- device-pointer calls (pointers are plain integers, e.g. torch.Tensor.data_ptr()) ----------
    def synth_frames_dev(self, d_gray, frames, H, W, first_frame, seed, cell, density_thr):
        self._chk(lib().orbamd_synth_frames_dev(self._h, _vp(d_gray), frames, H, W, first_frame,
                                                ctypes.c_uint32(seed & 0xFFFFFFFF), cell, density_thr),
                  "synth_frames_dev")

    def synth_textured_dev(self, d_gray, frames, H, W, first_frame, seed):
        self._chk(lib().orbamd_synth_textured_dev(self._h, _vp(d_gray), frames, H, W, first_frame,
                                                  ctypes.c_uint32(seed & 0xFFFFFFFF)), "synth_textured_dev")

    def bgr2gray_dev(self, d_bgr, frames, H, W, d_gray):
        self._chk(lib().orbamd_bgr2gray_dev(self._h, _vp(d_bgr), frames, H, W, _vp(d_gray)), "bgr2gray_dev")

    def extract_batch_dev(self, d_gray, frames, H, W, cap, d_kp7, d_desc, d_valid_idx, d_counts):
        self._chk(lib().orbamd_extract_batch_dev(self._h, _vp(d_gray), frames, H, W, cap, _vp(d_kp7), _vp(d_desc),
                                                 _vp(d_valid_idx), _vp(d_counts)), "extract_batch_dev")

    def knn2_batch_dev(self, d_desc, d_counts, frames, cap, d_pairs, npairs, d_idx, d_dist, d_good):
        self._chk(lib().orbamd_knn2_batch_dev(self._h, _vp(d_desc), _vp(d_counts), frames, cap, _vp(d_pairs or 0),
                                              npairs, _vp(d_idx), _vp(d_dist), _vp(d_good or 0)), "knn2_batch_dev")

    def ratio_filter_batch_dev(self, d_idx, d_good, d_counts, frames, cap, d_pairs, npairs, d_good_q, d_good_t2,
                               d_good_t1, d_kept):
        self._chk(lib().orbamd_ratio_filter_batch_dev(self._h, _vp(d_idx), _vp(d_good), _vp(d_counts), frames, cap,
                                                      _vp(d_pairs or 0), npairs, _vp(d_good_q), _vp(d_good_t2),
                                                      _vp(d_good_t1), _vp(d_kept)), "ratio_filter_batch_dev")

    def knn2_cross_dev(self, d_qdesc, d_qcounts, qframe, d_tdesc, d_tcounts, tframe, cap, d_idx, d_dist, d_good=0):
        self._chk(lib().orbamd_knn2_cross_dev(self._h, _vp(d_qdesc), _vp(d_qcounts), qframe, _vp(d_tdesc),
                                              _vp(d_tcounts), tframe, cap, _vp(d_idx), _vp(d_dist),
                                              _vp(d_good or 0)), "knn2_cross_dev")

    def knn2_dev(self, d_q, nq, d_t, nt, d_idx, d_dist, d_good=0):
        self._chk(lib().orbamd_knn2_dev(self._h, _vp(d_q), nq, _vp(d_t or 0), nt, _vp(d_idx), _vp(d_dist),
                                        _vp(d_good or 0)), "knn2_dev")

    def trig_selftest_dev(self, d_m01, d_m10, n, d_deg, d_unsure, d_rad, m, d_sin, d_cos):
        """Test hook: the orientation step's own atan2 / sincos, element-wise on the device."""
        self._chk(lib().orbamd_trig_selftest_dev(self._h, _vp(d_m01 or 0), _vp(d_m10 or 0), n, _vp(d_deg or 0),
                                                 _vp(d_unsure or 0), _vp(d_rad or 0), m, _vp(d_sin or 0),
                                                 _vp(d_cos or 0)), "trig_selftest_dev")

    def knn2_dev_noalloc(self, d_q, nq, d_t, nt, d_idx, d_dist, d_good=0):
        """knn2_dev on the library's fixed scratch array: no allocation, nq <= NOALLOC_MAX_QUERIES."""
        self._chk(lib().orbamd_knn2_dev_noalloc(self._h, _vp(d_q), nq, _vp(d_t or 0), nt, _vp(d_idx), _vp(d_dist),
                                                _vp(d_good or 0)), "knn2_dev_noalloc")
