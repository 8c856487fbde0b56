"""Seeded synthetic inputs for tests and benchmarks (host-side NumPy definition).

The frame generator is a pure function of (seed, frame index, pixel): every value comes from a
32-bit integer hash, so the same frames can be produced bit-for-bit by NumPy here and by the
``synth_frames`` HIP kernel on the device (``orbamd_synth_frames_dev``), with no host-to-device
copy of gigabytes of pixels.

Scene: a static "world" scrolls under the camera by one pixel per frame along x.  The background
is world-anchored noise in [96, 104] (spread 8 < FAST threshold 20, so it makes no corners by
itself) with a sparse per-frame flicker of +-1.  The world is cut into ``cell`` x ``cell`` squares;
a square holds, with probability ``density``, one blob: a single pixel or a 2x2 block at 230 or
10, placed at least 4 px from the square's edge (blobs are >= 9 px apart).  A lone pixel yields
exactly one FAST keypoint and a 2x2 block four, so the keypoint count is about
``2.5 * density * H * W / cell**2``.  Every square re-draws its blob every 20 frames with a
per-square phase, i.e. 5 % of the blobs change per frame.  Blobs drift through the 3..15 px border
band, so some keypoints have no descriptor.
"""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)

# Presets: (cell, density) giving the keypoint counts the workloads are quoted on.
PRESETS = {
    "480p": dict(H=480, W=640, cell=12, density=0.386),     # ~2k keypoints
    "1080p": dict(H=1080, W=1920, cell=20, density=0.386),  # ~5k keypoints
    "4k": dict(H=2160, W=3840, cell=20, density=0.386),     # ~20k keypoints
}
DEFAULT_SEED = 0xC0FFEE
LIFETIME = 20


def _mix(x):
    """lowbias32 integer finaliser on uint64 arrays holding 32-bit values."""
    x = x & M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & M32
    x ^= x >> np.uint64(16)
    return x


def hash3(seed, a, b, c):
    """h(seed, a, b, c) -> uint32 value in a uint64 array; a, b, c are non-negative integers."""
    a = np.asarray(a, np.uint64)
    b = np.asarray(b, np.uint64)
    c = np.asarray(c, np.uint64)
    h = _mix((np.uint64(seed & 0xFFFFFFFF) + np.uint64(0x9E3779B9) * (a & M32)) & M32)
    h = _mix(h ^ ((np.uint64(0x85EBCA6B) * (b & M32)) & M32))
    h = _mix((h + np.uint64(0xC2B2AE35) * (c & M32)) & M32)
    return h


def density_threshold(density):
    """Blob present iff (hash & 0xFFFF) < this."""
    return int(round(float(density) * 65536.0))


def synth_frame(H, W, frame_index, seed=DEFAULT_SEED, cell=20, density=0.386):
    """One gray frame, uint8 [H, W]."""
    f = int(frame_index)
    y, x = np.meshgrid(np.arange(H, dtype=np.uint64), np.arange(W, dtype=np.uint64), indexing="ij")
    wx = x + np.uint64(f)
    wy = y
    C = np.uint64(cell)
    cx, lx = wx // C, wx % C
    cy, ly = wy // C, wy % C
    phase = hash3(seed ^ 0x5EED, cx, cy, 7) % np.uint64(LIFETIME)
    epoch = (np.uint64(f) + phase) // np.uint64(LIFETIME)
    hc = hash3(seed ^ 0xB10B, cx, cy, epoch)
    has = (hc & np.uint64(0xFFFF)) < np.uint64(density_threshold(density))
    span = np.uint64(cell - 9)
    bx = np.uint64(4) + ((hc >> np.uint64(16)) & np.uint64(0xFF)) % span
    by = np.uint64(4) + ((hc >> np.uint64(24)) & np.uint64(0xFF)) % span
    h2 = _mix(hc ^ np.uint64(0xA5A5A5A5))
    size = np.uint64(1) + (h2 & np.uint64(1))
    bright = ((h2 >> np.uint64(1)) & np.uint64(1)) == 1
    inblob = has & (lx >= bx) & (lx < bx + size) & (ly >= by) & (ly < by + size)
    bg = np.uint64(96) + hash3(seed ^ 0xBAC6, wx, wy, 0) % np.uint64(9)
    hf = hash3(seed ^ 0xF11C, x, y, f)
    flick = (hf & np.uint64(15)) == 0
    bg = np.where(flick, np.where(bg < 104, bg + np.uint64(1), bg - np.uint64(1)), bg)
    out = np.where(inblob, np.where(bright, np.uint64(230), np.uint64(10)), bg)
    return out.astype(np.uint8)


# Textured scene: the same scrolling world, but every pixel belongs to a piecewise-constant texture
# instead of a flat background -- 16x16 blocks of a random level in 0 .. 127, plus 7x7 blocks of a
# random offset in 0 .. 80, plus per-pixel noise in 0 .. 8, plus the +-1 flicker.  Local contrast is
# far above the FAST threshold everywhere, so the cheap screens reject little, corners sit on block
# corners in clusters of adjacent pixels, and a 1080p frame holds about 11 k keypoints.
TEX_COARSE_SHIFT, TEX_MID_CELL, TEX_MID_LEVELS, TEX_NOISE_LEVELS = 4, 7, 81, 9


def synth_textured_frame(H, W, frame_index, seed=DEFAULT_SEED):
    """One gray frame of the textured scene, uint8 [H, W]."""
    f = int(frame_index)
    y, x = np.meshgrid(np.arange(H, dtype=np.uint64), np.arange(W, dtype=np.uint64), indexing="ij")
    wx = x + np.uint64(f)
    wy = y
    a = hash3(seed ^ 0x7E01, wx >> np.uint64(TEX_COARSE_SHIFT), wy >> np.uint64(TEX_COARSE_SHIFT), 0) & np.uint64(255)
    b = hash3(seed ^ 0x7E02, wx // np.uint64(TEX_MID_CELL), wy // np.uint64(TEX_MID_CELL), 0) % np.uint64(TEX_MID_LEVELS)
    n = hash3(seed ^ 0x7E03, wx, wy, 0) % np.uint64(TEX_NOISE_LEVELS)
    v = np.uint64(30) + (a >> np.uint64(1)) + b + n          # 30 .. 245
    hf = hash3(seed ^ 0xF11C, x, y, f)
    flick = (hf & np.uint64(15)) == 0
    v = np.where(flick, np.where((hf >> np.uint64(4)) & np.uint64(1) == 1, v + np.uint64(1), v - np.uint64(1)), v)
    return v.astype(np.uint8)


def synth_textured_frames(H, W, first_frame, count, seed=DEFAULT_SEED):
    return np.stack([synth_textured_frame(H, W, first_frame + i, seed) for i in range(count)])


def synth_frames(H, W, first_frame, count, seed=DEFAULT_SEED, cell=20, density=0.386):
    return np.stack([synth_frame(H, W, first_frame + i, seed, cell, density) for i in range(count)])


def synth_descriptors(n, seed, dup_frac=0.01):
    """[n, 32] uint8 i.i.d. uniform bytes with ``dup_frac`` of the rows overwritten by copies of
    other rows, so that exact ties in distance occur (dense matching stress input)."""
    rng = np.random.default_rng(seed)
    d = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    ndup = int(n * dup_frac)
    if ndup:
        dst = rng.choice(n, size=ndup, replace=False)
        src = rng.integers(0, n, size=ndup)
        d[dst] = d[src]
    return d
