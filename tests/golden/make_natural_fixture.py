"""Builds tests/golden/natural_pair.npz: the two halves of a photograph the reference ships as
documentation (images/"Screenshot 2025-11-10 at 12.03.47.png": two camera views of a corridor side
by side, an upscaled screen capture, so every fourth pixel is taken), 392x440 BGR each, decoded with
Pillow in the build container, plus what the CPU restatement (oracle/) gives for them.  The crops are data (pixels); the expected outputs come from the restatement, not from
the reference, so they pin the GPU library to the restatement on natural texture, nothing more.

    python tests/golden/make_natural_fixture.py            # needs /root/reference and Pillow
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

SRC = "/root/reference/images/Screenshot 2025-11-10 at 12.03.47.png"
STEP = 4
CROPS = {"a": (0, 0), "b": (440, 0)}      # (x0, y0) of the left and the right view after decimation
H, W = 392, 440


def main():
    from PIL import Image
    import oracle
    rgb = np.asarray(Image.open(SRC).convert("RGB"))[::STEP, ::STEP]
    out = {}
    for name, (x0, y0) in CROPS.items():
        bgr = np.ascontiguousarray(rgb[y0:y0 + H, x0:x0 + W, ::-1])
        kp, desc = oracle.c.extract(bgr)
        out["bgr_" + name] = bgr
        out["n_" + name] = np.int64(len(kp))
        out["nv_" + name] = np.int64(len(desc))
        out["kp_sha_" + name] = np.frombuffer(hashlib.sha256(kp.tobytes()).digest(), np.uint8)
        out["desc_sha_" + name] = np.frombuffer(hashlib.sha256(desc.tobytes()).digest(), np.uint8)
        out["desc_" + name] = desc
    idx, dist = oracle.c.knn2(out["desc_b"], out["desc_a"])
    out["match_sha"] = np.frombuffer(hashlib.sha256(idx.tobytes() + dist.tobytes()).digest(), np.uint8)
    out["good"] = oracle.c.ratio_filter(idx, dist)[0]
    del out["desc_a"], out["desc_b"]
    np.savez_compressed(os.path.join(HERE, "natural_pair.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") and v.ndim else int(v)) for k, v in out.items()})


if __name__ == "__main__":
    main()
