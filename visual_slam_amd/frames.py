"""Frame sources that need no codec: YUV4MPEG2 (.y4m) and headerless raw gray files.

The reference opens its input with cv.VideoCapture(path) and pulls one frame per iteration with
cap.read() (reference main.py:80, views.py:81).  Neither OpenCV nor any decoder library is part of this
package; what it can read by itself are uncompressed streams, which any decoder produces
(`ffmpeg -i video.mp4 -pix_fmt gray video.y4m`).  FrameFile mirrors the part of cv.VideoCapture the reference
uses -- isOpened(), read() -> (ok, frame), release() -- and adds read_into(), which lets the frame stager
(visual_slam_amd.staging.HostFrameStager.run_source) read straight into its pinned host chunks.

Frames come out gray [H, W] uint8: the luma plane of a Y4M stream, or the bytes of a raw file.  With
bgr=True they come out as [H, W, 3] with the three channels equal, the shape cap.read() returns and
orb.extract() takes (reference src/main_interface.cpp:80-83); the BGR -> gray conversion of such a frame gives
back the luma values exactly.
"""
import os

import numpy as np

_Y4M_MAGIC = b"YUV4MPEG2"


def _chroma_bytes(colourspace, W, H):
    """Bytes after the luma plane in one frame of an 8-bit Y4M stream."""
    cs = colourspace
    cw, ch = (W + 1) // 2, (H + 1) // 2
    if cs.startswith("mono"):
        extra = 0
    elif cs.startswith("420"):
        extra = 2 * cw * ch
    elif cs.startswith("422"):
        extra = 2 * cw * H
    elif cs.startswith("411"):
        extra = 2 * ((W + 3) // 4) * H
    elif cs == "444alpha":
        extra = 3 * W * H
    elif cs.startswith("444"):
        extra = 2 * W * H
    else:
        raise ValueError("Y4M colour space %r is not supported" % cs)
    for depth in ("p9", "p10", "p12", "p14", "p16", "mono9", "mono10", "mono12", "mono16"):
        if cs.endswith(depth):
            raise ValueError("Y4M colour space %r: only 8-bit samples are supported" % cs)
    return extra


class FrameFile:
    """Sequential reader of a .y4m file, or of a raw gray file when H and W are given."""

    def __init__(self, path, H=None, W=None, bgr=False):
        self.path, self.bgr = path, bool(bgr)
        self._f = open(path, "rb")
        self.fps = None
        self.colourspace = "mono"
        self._y4m = False
        head = self._f.read(len(_Y4M_MAGIC))
        if head == _Y4M_MAGIC:
            self._y4m = True
            line = self._f.readline(4096)
            if not line.endswith(b"\n"):
                self._f.close()
                raise ValueError("%s: Y4M stream header is not terminated" % path)
            w = h = None
            cs = "420jpeg"
            for tok in line.split():
                tag, val = chr(tok[0]), tok[1:].decode("ascii", "replace")
                if tag == "W":
                    w = int(val)
                elif tag == "H":
                    h = int(val)
                elif tag == "C":
                    cs = val
                elif tag == "F" and ":" in val:
                    num, den = val.split(":")
                    self.fps = float(num) / float(den) if float(den) else None
            if not w or not h or w <= 0 or h <= 0:
                self._f.close()
                raise ValueError("%s: Y4M header without a frame size" % path)
            if (H and H != h) or (W and W != w):
                self._f.close()
                raise ValueError("%s: stream is %dx%d, not the %sx%s asked for" % (path, w, h, W, H))
            self.H, self.W, self.colourspace = h, w, cs
            try:
                self._extra = _chroma_bytes(cs, w, h)
            except ValueError:
                self._f.close()
                raise
        else:
            if not H or not W:
                self._f.close()
                raise ValueError("%s is not a Y4M stream; a raw gray file needs H and W" % path)
            self.H, self.W, self._extra = int(H), int(W), 0
            self._f.seek(0)
        self._luma = self.H * self.W
        self.index = 0          # frames handed out so far

    # -- the part of cv.VideoCapture the reference uses ---------------------------------------------
    def isOpened(self):
        return self._f is not None and not self._f.closed

    def release(self):
        if self._f is not None:
            self._f.close()

    def read(self):
        """-> (True, frame) or (False, None) at the end of the stream (or on a truncated last frame)."""
        out = np.empty((self.H, self.W, 3) if self.bgr else (self.H, self.W), np.uint8)
        if not self.read_into(out):
            return False, None
        return True, out

    # -- streaming --------------------------------------------------------------------------------------
    def read_into(self, out):
        """Next frame into `out` (uint8, C-contiguous, [H, W] or [H, W, 3]).  False at the end of the stream."""
        if not self.isOpened():
            return False
        if out.dtype != np.uint8 or not out.flags.c_contiguous or out.shape[:2] != (self.H, self.W) or \
                (out.ndim == 3 and out.shape[2] != 3) or out.ndim not in (2, 3):
            raise ValueError("frame buffer must be C-contiguous uint8 [%d, %d] or [%d, %d, 3]" % (self.H, self.W, self.H, self.W))
        if self._y4m:
            line = self._f.readline(4096)
            if not line:
                return False
            if not line.startswith(b"FRAME") or not line.endswith(b"\n"):
                raise ValueError("%s: frame %d does not start with a FRAME header" % (self.path, self.index))
        if out.ndim == 2:
            got = self._f.readinto(memoryview(out).cast("B"))
        else:
            luma = np.empty((self.H, self.W), np.uint8)
            got = self._f.readinto(memoryview(luma).cast("B"))
            out[...] = luma[:, :, None]
        if got != self._luma:
            return False                    # truncated last frame
        if self._extra:
            self._f.seek(self._extra, os.SEEK_CUR)
        self.index += 1
        return True

    def __iter__(self):
        while True:
            ok, frame = self.read()
            if not ok:
                return
            yield frame

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.release()
        return False


def VideoCapture(path, H=None, W=None):
    """cv.VideoCapture(path) for a file this package can read: frames come out [H, W, 3] like cap.read()."""
    return FrameFile(path, H=H, W=W, bgr=True)


def write_y4m(path, frames, fps=30, colourspace="mono"):
    """Write gray frames (iterable of uint8 [H, W]) as a Y4M stream; colourspace "mono" or "420jpeg" (neutral chroma)."""
    if colourspace not in ("mono", "420jpeg"):
        raise ValueError("write_y4m writes mono or 420jpeg")
    n = 0
    with open(path, "wb") as f:
        for fr in frames:
            fr = np.ascontiguousarray(fr)
            if fr.dtype != np.uint8 or fr.ndim != 2:
                raise ValueError("frames must be uint8 [H, W]")
            H, W = fr.shape
            if n == 0:
                f.write(b"YUV4MPEG2 W%d H%d F%d:1 Ip A1:1 C%s\n" % (W, H, int(fps), colourspace.encode()))
                shape = (H, W)
                chroma = bytes([128]) * (2 * ((W + 1) // 2) * ((H + 1) // 2)) if colourspace == "420jpeg" else b""
            elif fr.shape != shape:
                raise ValueError("all frames must have one size")
            f.write(b"FRAME\n")
            f.write(fr.tobytes())
            f.write(chroma)
            n += 1
    return n


def write_raw_gray(path, frames):
    n = 0
    with open(path, "wb") as f:
        for fr in frames:
            fr = np.ascontiguousarray(fr)
            if fr.dtype != np.uint8 or fr.ndim != 2:
                raise ValueError("frames must be uint8 [H, W]")
            f.write(fr.tobytes())
            n += 1
    return n
