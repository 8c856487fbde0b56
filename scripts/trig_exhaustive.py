#!/usr/bin/env python3
"""Every float angle in [0, 360]: sine / cosine of visual_slam_amd/csrc/orb_trig.h (host build, same arithmetic as
the device) against this machine's libm.  Prints how many differ and the largest difference in units of the last
place.  About a minute on 8 threads.

    python scripts/trig_exhaustive.py [threads]
"""
import ctypes
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(ROOT, "build", "libtrig_host.so")
subprocess.check_call(["g++", "-O2", "-mfma", "-ffp-contract=off", "-shared", "-fPIC", "-o", so,
                       os.path.join(ROOT, "tests", "trig_host.cpp"), "-lm"])
L = ctypes.CDLL(so)
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hi = int(np.float32(360.0).view(np.uint32)) + 1
edges = np.linspace(0, hi, 64 * threads + 1).astype(np.int64)


def run(i):
    out = (ctypes.c_long * 4)()
    L.trig_sincos_vs_libm(ctypes.c_uint32(int(edges[i])), ctypes.c_uint32(int(edges[i + 1])), out)
    return list(out)


with ThreadPoolExecutor(threads) as ex:
    res = list(ex.map(run, range(len(edges) - 1)))
n = sum(r[0] for r in res)
ds = sum(r[1] for r in res)
dc = sum(r[2] for r in res)
worst = max(r[3] for r in res)
print("float angles in [0, 360]: %d" % n)
print("sines that differ from libm: %d (%.3f %%)   cosines: %d (%.3f %%)" % (ds, 100.0 * ds / n, dc, 100.0 * dc / n))
print("largest difference: %d unit(s) in the last place" % worst)
