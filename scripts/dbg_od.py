import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
import oracle
from conftest import make_test_image
from visual_slam_amd._lib import Context
c = Context(0)
from visual_slam_amd import synth
P = synth.PRESETS["480p"]
for fr in (0, 1):
    img = synth.synth_frame(P["H"], P["W"], fr, cell=P["cell"], density=P["density"])
    H, W = img.shape
    kp, desc, vidx = c.extract_gray(img)
    ekp, edesc, evidx = oracle.c.extract_gray(img)
    print(H, W, "n", len(kp), len(ekp), "nv", len(desc), len(edesc))
    if len(kp) == len(ekp):
        badk = np.nonzero((kp.view(np.uint32) != ekp.view(np.uint32)).any(1))[0]
        print("  kp mismatches", len(badk))
        for b in badk[:5]:
            print("   ", b, kp[b], ekp[b])
    if desc.shape == edesc.shape:
        bad = np.nonzero((desc != edesc).any(1))[0]
        print("  desc mismatches", len(bad), "of", len(desc))
        for b in bad[:8]:
            k = vidx[b]
            diff = np.unpackbits(desc[b] ^ edesc[b], bitorder="little")
            print("   row", b, "kp", ekp[k][:4], "bits differing", int(diff.sum()), np.nonzero(diff)[0][:10])
