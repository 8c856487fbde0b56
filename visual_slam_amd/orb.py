"""Host-side mirror of the reference's ORB interface, backed by liborbamd.so.

Names and argument meaning follow the reference so that its call sites read the same:

  ORBDescriptor.detectKeypoints / computeOrientation / computeDescriptors
      reference include/orb.hpp:14-20 (static methods of ORBDescriptor)
  extract(image)
      reference src/main_interface.cpp:80-101 (the pybind entry point views.py:273 calls)
  BFMatcher(...).knnMatch(d1, d2, 2)
      cv2.BFMatcher(cv2.NORM_HAMMING, crossCheck=False).knnMatch at reference views.py:65, :162
  lowe_ratio_matches(matches) / good_matches(d1, d2)
      the filter loop of reference views.py:163-171

Keypoints travel as float32 rows [x, y, size, angle, response, octave, class_id], the layout the
reference's module returns (src/main_interface.cpp:39-55).  Everything computes on the GPU; with
no GPU every call raises.
"""
from collections import namedtuple

import numpy as np

from . import _lib

NORM_HAMMING = 6  # value of cv2.NORM_HAMMING, accepted for call-site compatibility

_ctx = None


def context(device=None):
    """Process-wide context (created on first use; ORBAMD_DEVICE selects the GPU)."""
    global _ctx
    if _ctx is None:
        import os
        dev = int(os.environ.get("ORBAMD_DEVICE", "0")) if device is None else device
        _ctx = _lib.Context(dev)
    return _ctx


def _gray_of(image):
    image = np.asarray(image)
    if image.dtype != np.uint8:
        raise RuntimeError("Image must be uint8")
    if image.ndim == 3 and image.shape[2] == 3:
        return context().bgr2gray(image)
    if image.ndim == 2:
        return np.ascontiguousarray(image)
    raise RuntimeError("Expect HxW (grey) or HxWx3 (BGR) image")


class ORBDescriptor:
    """The three stages the reference exposes as static methods (include/orb.hpp:8-26)."""

    @staticmethod
    def detectKeypoints(image):
        """FAST segment test (reference src/orb.cpp:24-108).  -> float32 [N, 7]; fields other than
        x, y, size hold cv::KeyPoint's defaults (angle -1, response 0, octave 0, class_id -1)."""
        gray = _gray_of(image)
        xy = context().detect(gray)
        kp = np.zeros((len(xy), 7), np.float32)
        kp[:, 0:2] = xy
        kp[:, 2] = 7.0
        kp[:, 3] = -1.0
        kp[:, 6] = -1.0
        return kp

    @staticmethod
    def computeOrientation(image, keypoints):
        """Writes the angle column in place (reference src/orb.cpp:110-140) and returns the array."""
        gray = _gray_of(image)
        xy = np.rint(keypoints[:, 0:2]).astype(np.int32)
        keypoints[:, 3] = context().orient(gray, xy)
        return keypoints

    @staticmethod
    def computeDescriptors(image, keypoints):
        """Steered BRIEF-256 (reference src/orb.cpp:142-234).  -> uint8 [Nv, 32] for the keypoints at
        least 16 px from every edge, in keypoint order."""
        gray = _gray_of(image)
        xy = np.rint(keypoints[:, 0:2]).astype(np.int32)
        desc, _ = context().describe(gray, xy, np.ascontiguousarray(keypoints[:, 3], np.float32))
        return desc


def extract(image):
    """extract(image) of the reference's module: BGR uint8 [H, W, 3] -> (kp [N,7], desc [Nv,32]);
    desc has shape (0, 0) when no keypoint is found (src/main_interface.cpp:89-98)."""
    image = np.asarray(image)
    if not (image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 3)):
        raise RuntimeError("Expect HxW (grey) or HxWx3 (BGR) image")
    if image.dtype != np.uint8:
        raise RuntimeError("Image must be uint8 or float32" if image.dtype != np.float32
                           else "float32 images are not supported by the ORB kernels; pass uint8")
    if image.ndim != 3:
        raise RuntimeError("extract expects an HxWx3 BGR image (BGR2GRAY needs 3 channels)")
    kp, desc, _ = context().extract_bgr(image)
    if len(kp) == 0:
        desc = np.empty((0, 0), np.uint8)
    return kp, desc


DMatch = namedtuple("DMatch", ["queryIdx", "trainIdx", "imgIdx", "distance"])


class BFMatcher:
    """Drop-in for the reference's use of cv2.BFMatcher (views.py:65): Hamming norm, no cross check."""

    def __init__(self, normType=NORM_HAMMING, crossCheck=False):
        if normType != NORM_HAMMING:
            raise ValueError("only NORM_HAMMING is implemented")
        if crossCheck:
            raise ValueError("crossCheck=True is not implemented (the reference passes False)")

    def knnMatchArrays(self, d1, d2):
        """-> (idx int32 [N,2], dist int32 [N,2]); -1 where the train set has too few rows."""
        d1 = np.asarray(d1, np.uint8).reshape(-1, 32) if np.size(d1) else np.zeros((0, 32), np.uint8)
        d2 = np.asarray(d2, np.uint8).reshape(-1, 32) if np.size(d2) else np.zeros((0, 32), np.uint8)
        return context().knn2(d1, d2)

    def knnMatch(self, d1, d2, k=2):
        """List (one entry per query row) of lists of up to k DMatch, best first, as OpenCV returns."""
        if k not in (1, 2):
            raise ValueError("k must be 1 or 2")
        idx, dist = self.knnMatchArrays(d1, d2)
        out = []
        for q in range(len(idx)):
            out.append([DMatch(q, int(idx[q, j]), 0, float(dist[q, j])) for j in range(k) if idx[q, j] >= 0])
        return out

    def match(self, d1, d2):
        """Best match per query (used by the reference only for drawing, views.py:147, :203)."""
        idx, dist = self.knnMatchArrays(d1, d2)
        return [DMatch(q, int(idx[q, 0]), 0, float(dist[q, 0])) for q in range(len(idx)) if idx[q, 0] >= 0]


def good_matches(d1, d2):
    """knnMatch + ratio 0.75 on the GPU.  -> (good_f1, good_f2, first_train): for each kept query,
    m1.queryIdx, m2.trainIdx (sic: the reference stores the SECOND neighbour, views.py:170) and
    m1.trainIdx (the index views.py:171 uses for the point pair)."""
    idx, dist = BFMatcher().knnMatchArrays(d1, d2)
    return context().ratio_filter(idx, dist)


def lowe_ratio_matches(matches, filter_scale=0.75):
    """The reference's Python loop (views.py:163-171) over knnMatch output, for callers that hold
    DMatch lists.  Raises ValueError, as tuple unpacking does, if a query has fewer than 2."""
    good_f1, good_f2, first = [], [], []
    for pair in matches:
        m1, m2 = pair
        if m1.distance < m2.distance * filter_scale:
            good_f1.append(m1.queryIdx)
            good_f2.append(m2.trainIdx)
            first.append(m1.trainIdx)
    return np.array(good_f1, np.int64), np.array(good_f2, np.int64), np.array(first, np.int64)
