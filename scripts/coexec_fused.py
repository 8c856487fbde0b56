#!/usr/bin/env python3
"""Developer tool: the orientation + descriptor kernel of one batch and the matcher of another batch as ONE
launch with role-specialised workgroups (both kinds resident on every compute unit at the same time), against
the two kernels launched one after the other.  Results are compared bit for bit.

    python3 scripts/coexec_fused.py [--frames 1000] [--reps 5] [--only separate|fused4|fused2]
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from visual_slam_amd import _lib, synth  # noqa: E402
from visual_slam_amd.engine import OrbEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=1000)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--only", default="")
args = ap.parse_args()
P = synth.PRESETS["1080p"]
F, H, W, cap = args.frames, P["H"], P["W"], 8192
eng = OrbEngine(0, cap=cap)
grayA = eng.synth_frames(F, H, W, first_frame=0, cell=P["cell"], density=P["density"])
grayB = eng.synth_frames(F, H, W, first_frame=5000, cell=P["cell"], density=P["density"])
featB = eng.extract_batch(grayB)
featA = eng.alloc_features(F)
featA2 = eng.alloc_features(F)
mB = eng.alloc_matches(F - 1)
mB2 = eng.alloc_matches(F - 1)
L = _lib.lib()
vp = ctypes.c_void_p
L.orbamd_x_describe_match_dev.restype = ctypes.c_int
L.orbamd_x_describe_match_dev.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp,
                                          vp, vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_int]


def fused(qt, fa, mb):
    st = L.orbamd_x_describe_match_dev(eng.ctx._h, vp(grayA.data_ptr()), F, H, W, cap, vp(fa.kp7.data_ptr()),
                                       vp(fa.desc.data_ptr()), vp(fa.valid_idx.data_ptr()), vp(fa.counts.data_ptr()),
                                       vp(featB.desc.data_ptr()), vp(featB.counts.data_ptr()), F, cap,
                                       vp(mb.idx.data_ptr()), vp(mb.dist.data_ptr()), vp(mb.good.data_ptr()), qt)
    assert st == 0, st


def separate(fa, mb):
    eng.extract_batch(grayA, out=fa)
    eng.match_consecutive(featB, out=mb)


def timed(fn, reps):
    fn()
    eng.sync()
    eng.ctx.timing_enable(True)
    eng.ctx.timing_reset()
    for _ in range(reps):
        fn()
    t = eng.ctx.timing_read()
    eng.ctx.timing_enable(False)
    return {k: v[0] / v[1] for k, v in t.items() if v[1]}


if args.only in ("", "separate"):
    t = timed(lambda: separate(featA, mB), args.reps)
    print("separate launches:  describe %.3f ms + knn2 %.3f ms = %.3f ms" %
          (t["orient_describe"], t["knn2"], t["orient_describe"] + t["knn2"]))
for qt in (4, 2):
    if args.only in ("", "fused%d" % qt):
        t = timed(lambda: fused(qt, featA2, mB2), args.reps)
        print("one launch, matcher role with %d query tiles per wave (%s workgroups per CU by registers): %.3f ms"
              % (qt, "3" if qt == 4 else "4", t["orient_describe"]))
        if args.only == "":
            eng.sync()
            n = featA.counts[:, 0].clamp(max=cap)
            ok = torch.equal(featA.counts, featA2.counts)
            ar = torch.arange(cap, device=eng.device)[None, :]
            live = (ar < n[:, None])
            ok &= bool(((featA.kp7.view(torch.int32) == featA2.kp7.view(torch.int32)).all(2) | ~live).all())
            livev = ar < featA.counts[:, 1].clamp(max=cap)[:, None]
            ok &= bool(((featA.desc == featA2.desc).all(2) | ~livev).all())
            liveq = ar < featB.counts[1:, 1].clamp(max=cap)[:, None]
            ok &= bool(((mB.idx == mB2.idx).all(2) | ~liveq).all()) and bool(((mB.dist == mB2.dist).all(2) | ~liveq).all())
            print("   results equal to the separate launches:", ok)
            assert ok
