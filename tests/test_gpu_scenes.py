"""GPU: content other than the sparse benchmark scene -- natural texture (a photograph), the textured
synthetic scene, and the full-size 1080p batch (1 000 frames) with properties checked on every frame
and the CPU restatement on a sample of them."""
import numpy as np
import pytest
import torch

import oracle
from test_natural_fixture import check_natural_pair
from visual_slam_amd import synth

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_natural_texture_fixture_through_the_module():
    import build.orb_project as orb
    check_natural_pair(orb.extract, orb.knn_match2, orb.ratio_filter)


@pytest.fixture(scope="module")
def eng16k():
    from visual_slam_amd.engine import OrbEngine
    return OrbEngine(0, cap=16384)


def test_textured_scene_device_equals_numpy(eng16k):
    for H, W, first, n in ((97, 131, 0, 2), (480, 640, 1234, 2), (1080, 1920, 5, 1)):
        g = eng16k.synth_textured_frames(n, H, W, first_frame=first)
        eng16k.sync()
        assert np.array_equal(g.cpu().numpy(), synth.synth_textured_frames(H, W, first, n))


def test_textured_1080p_frames_equal_the_restatement(eng16k):
    """About 11 k keypoints per frame in clusters, local contrast above the FAST threshold everywhere:
    the detector's cheap screens reject little and the descriptor windows are shared differently."""
    eng = eng16k
    H, W, F = 1080, 1920, 3
    gray = eng.synth_textured_frames(F, H, W, first_frame=40)
    feat = eng.extract_batch(gray)
    m = eng.match_consecutive(feat)
    eng.sync()
    assert feat.overflowed() == []
    host = gray.cpu().numpy()
    ex = [oracle.c.extract_gray(f) for f in host]
    assert min(len(e[0]) for e in ex) > 9000
    for f in range(F):
        kp, desc, vidx = feat.frame(f)
        assert np.array_equal(bits(kp), bits(ex[f][0]))
        assert np.array_equal(desc, ex[f][1]) and np.array_equal(vidx, ex[f][2])
    for f in range(1, F):
        eidx, edist = oracle.c.knn2(ex[f][1], ex[f - 1][1])
        idx, dist, good = m.pair(f - 1, len(ex[f][1]))
        assert np.array_equal(idx, eidx) and np.array_equal(dist, edist)
        assert np.array_equal(np.nonzero(good)[0], oracle.c.ratio_filter(eidx, edist)[0])


@pytest.mark.timeout(600)
def test_full_size_1080p_batch_properties_and_sampled_restatement():
    """The batch the headline figure is quoted on: 1 000 frames of 1920x1080, about 5 k keypoints each.
    Every frame: scan order, row format, angle range, descriptor/keypoint linkage, neighbour order and
    ratio flags (checked on the device).  Sample: the first and last frames and the frames on either
    side of each eighth of the batch (where the kernels' work-item remapping hands over from one XCD
    to the next), compared bit for bit with the CPU restatement."""
    from visual_slam_amd.engine import OrbEngine
    P = synth.PRESETS["1080p"]
    H, W, F, cap = P["H"], P["W"], 1000, 8192
    eng = OrbEngine(0, cap=cap)
    gray = eng.synth_frames(F, H, W, first_frame=0, cell=P["cell"], density=P["density"])
    feat = eng.extract_batch(gray)
    m = eng.match_consecutive(feat)
    eng.sync()
    counts = feat.counts.long()
    n, nv = counts[:, 0], counts[:, 1]
    assert int(n.max()) <= cap and int(n.min()) > 3000 and bool((nv <= n).all())
    ar = torch.arange(cap, device=eng.device)[None, :]
    live = ar < n[:, None]                                   # [F, cap]
    x, y = feat.kp7[:, :, 0], feat.kp7[:, :, 1]
    key = (y.long() * W + x.long())
    # scan order: strictly ascending (y, x) inside every frame
    inc = key[:, 1:] > key[:, :-1]
    assert bool((inc | ~live[:, 1:]).all())
    assert bool(((x >= 3) & (x < W - 3) & (y >= 3) & (y < H - 3) | ~live).all())
    for col, val in ((2, 7.0), (4, 0.0), (5, 0.0), (6, -1.0)):
        assert bool(((feat.kp7[:, :, col] == val) | ~live).all())
    ang = feat.kp7[:, :, 3]
    assert bool((((ang >= 0) & (ang <= 360)) | ~live).all())
    # descriptor rows: valid_idx ascending, pointing at keypoints >= 16 px from every edge, all of them
    livev = ar < nv[:, None]
    vi = feat.valid_idx.long().clamp(0, cap - 1)
    assert bool(((vi[:, 1:] > vi[:, :-1]) | ~livev[:, 1:]).all())
    vx, vy = torch.gather(x, 1, vi), torch.gather(y, 1, vi)
    assert bool((((vx >= 16) & (vx < W - 16) & (vy >= 16) & (vy < H - 16)) | ~livev).all())
    inner = (x >= 16) & (x < W - 16) & (y >= 16) & (y < H - 16) & live
    assert torch.equal(inner.sum(1), nv)
    # matches: neighbours in (distance, index) order, indices inside the train frame, flags = ratio test
    nq, nt = nv[1:], nv[:-1]
    liveq = ar < nq[:, None]
    i1, i2 = m.idx[:, :, 0].long(), m.idx[:, :, 1].long()
    d1, d2 = m.dist[:, :, 0].long(), m.dist[:, :, 1].long()
    assert bool((((i1 >= 0) & (i1 < nt[:, None]) & (i2 >= 0) & (i2 < nt[:, None]) & (i1 != i2)) | ~liveq).all())
    assert bool((((d1 < d2) | ((d1 == d2) & (i1 < i2))) | ~liveq).all())
    assert bool((((d1 >= 0) & (d2 <= 256)) | ~liveq).all())
    assert bool(((m.good.bool() == (4 * d1 < 3 * d2)) | ~liveq).all())
    # the reported best distance is the true distance of the reported pair (popcount on the device)
    qd = feat.desc[1:]                                        # [F-1, cap, 32]
    t1 = torch.gather(feat.desc[:-1], 1, i1.clamp(0, cap - 1)[:, :, None].expand(-1, -1, 32))
    xor = (qd ^ t1).to(torch.int16)
    pop = torch.zeros_like(d1)
    for b in range(8):
        pop += ((xor >> b) & 1).sum(2)
    assert bool(((pop == d1) | ~liveq).all())

    sample = sorted({0, 1, F - 2, F - 1} | {k * F // 8 + o for k in range(1, 8) for o in (-1, 0)})
    need = sorted(set(sample) | {f - 1 for f in sample if f > 0})
    host = {f: gray[f].cpu().numpy() for f in need}
    ex = {f: oracle.c.extract_gray(host[f]) for f in need}
    for f in sample:
        assert np.array_equal(host[f], synth.synth_frame(H, W, f, cell=P["cell"], density=P["density"]))
        kp, desc, vidx = feat.frame(f)
        assert np.array_equal(bits(kp), bits(ex[f][0]))
        assert np.array_equal(desc, ex[f][1]) and np.array_equal(vidx, ex[f][2])
        if f > 0:
            eidx, edist = oracle.c.knn2(ex[f][1], ex[f - 1][1])
            idx, dist, _ = m.pair(f - 1, len(ex[f][1]))
            assert np.array_equal(idx, eidx) and np.array_equal(dist, edist)
