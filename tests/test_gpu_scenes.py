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


def check_batch_properties(eng, feat, m, H, W, cap, min_keypoints):
    """Size-independent properties of one extracted and matched batch, checked on the device for every frame:
    scan order, row format, angle range, descriptor / keypoint linkage, neighbour order, ratio flags, and the
    reported best distance against a popcount of the reported pair."""
    counts = feat.counts.long()
    n, nv = counts[:, 0], counts[:, 1]
    assert int(n.max()) <= cap and int(n.min()) > min_keypoints and bool((nv <= n).all())
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
    del key, inc, vx, vy, inner, vi
    # matches: neighbours in (distance, index) order, indices inside the train frame, flags = ratio test
    nq, nt = nv[1:], nv[:-1]
    liveq = ar < nq[:, None]
    i1, i2 = m.idx[:, :, 0].long(), m.idx[:, :, 1].long()
    d1, d2 = m.dist[:, :, 0].long(), m.dist[:, :, 1].long()
    assert bool((((i1 >= 0) & (i1 < nt[:, None]) & (i2 >= 0) & (i2 < nt[:, None]) & (i1 != i2)) | ~liveq).all())
    assert bool((((d1 < d2) | ((d1 == d2) & (i1 < i2))) | ~liveq).all())
    assert bool((((d1 >= 0) & (d2 <= 256)) | ~liveq).all())
    assert bool(((m.good.bool() == (4 * d1 < 3 * d2)) | ~liveq).all())
    # the reported best distance is the true distance of the reported pair (popcount on the device),
    # a slab of pairs at a time to bound the temporaries
    F = feat.frames
    for p0 in range(0, F - 1, 64):
        p1 = min(F - 1, p0 + 64)
        qd = feat.desc[p0 + 1:p1 + 1]                         # [pairs, cap, 32]
        t1 = torch.gather(feat.desc[p0:p1], 1, i1[p0:p1].clamp(0, cap - 1)[:, :, None].expand(-1, -1, 32))
        xor = (qd ^ t1).to(torch.int16)
        pop = torch.zeros_like(d1[p0:p1])
        for b in range(8):
            pop += ((xor >> b) & 1).sum(2)
        assert bool(((pop == d1[p0:p1]) | ~liveq[p0:p1]).all())


def check_sample_against_restatement(gray, feat, m, sample, regenerate):
    """Frames `sample` (and the pairs that end in them) bit for bit against the CPU restatement."""
    need = sorted(set(sample) | {f - 1 for f in sample if f > 0})
    host = {f: gray[f].cpu().numpy() for f in need}
    ex = {f: oracle.c.extract_gray(host[f]) for f in need}
    for f in sample:
        assert np.array_equal(host[f], regenerate(f))
        kp, desc, vidx = feat.frame(f)
        assert np.array_equal(bits(kp), bits(ex[f][0]))
        assert np.array_equal(desc, ex[f][1]) and np.array_equal(vidx, ex[f][2])
        if f > 0:
            eidx, edist = oracle.c.knn2(ex[f][1], ex[f - 1][1])
            idx, dist, _ = m.pair(f - 1, len(ex[f][1]))
            assert np.array_equal(idx, eidx) and np.array_equal(dist, edist)


@pytest.mark.timeout(600)
def test_full_size_1080p_batch_properties_and_sampled_restatement():
    """The batch the headline figure is quoted on: 1 000 frames of 1920x1080, about 5 k keypoints each.
    Every frame: the properties of check_batch_properties.  Sample: the first and last frames and the frames
    on either side of each eighth of the batch (where the kernels' work-item remapping hands over from one XCD
    to the next), compared bit for bit with the CPU restatement."""
    from visual_slam_amd.engine import OrbEngine
    P = synth.PRESETS["1080p"]
    H, W, F, cap = P["H"], P["W"], 1000, 8192
    eng = OrbEngine(0, cap=cap)
    gray = eng.synth_frames(F, H, W, first_frame=0, cell=P["cell"], density=P["density"])
    feat = eng.extract_batch(gray)
    m = eng.match_consecutive(feat)
    eng.sync()
    check_batch_properties(eng, feat, m, H, W, cap, 3000)
    sample = sorted({0, 1, F - 2, F - 1} | {k * F // 8 + o for k in range(1, 8) for o in (-1, 0)})
    check_sample_against_restatement(gray, feat, m, sample,
                                     lambda f: synth.synth_frame(H, W, f, cell=P["cell"], density=P["density"]))


@pytest.mark.timeout(900)
def test_one_rank_share_of_the_sharded_4k_batch():
    """One GPU's share of the 4 000-frame 3840x2160 batch cut over 8 GPUs: frames 1500 .. 1999 of the scene
    (rank 3's range), about 20 k keypoints each, capacity 32 768.  Every frame through the property checks,
    a sample (both ends of the range and its middle) bit for bit against the CPU restatement."""
    from visual_slam_amd.engine import OrbEngine
    from visual_slam_amd.scheduler import partition_frames
    P = synth.PRESETS["4k"]
    H, W, cap = P["H"], P["W"], 32768
    first, last = partition_frames(4000, 8)[3]
    F = last - first
    assert F == 500
    eng = OrbEngine(0, cap=cap)
    gray = eng.synth_frames(F, H, W, first_frame=first, cell=P["cell"], density=P["density"])
    feat = eng.extract_batch(gray)
    m = eng.match_consecutive(feat)
    eng.sync()
    assert feat.overflowed() == []
    check_batch_properties(eng, feat, m, H, W, cap, 15000)
    check_sample_against_restatement(gray, feat, m, [0, 1, F // 2, F - 1],
                                     lambda f: synth.synth_frame(H, W, first + f, cell=P["cell"], density=P["density"]))
