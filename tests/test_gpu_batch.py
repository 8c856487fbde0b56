"""GPU: frame-batch (device-pointer) entry points, golden fixtures, full-size property checks and the
device-libm parity gate.  Everything goes through the C ABI (ctypes) or the pybind module."""
import numpy as np
import pytest

import oracle
from conftest import make_test_image
from visual_slam_amd import _lib, synth

import golden_util

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def eng():
    from visual_slam_amd.engine import OrbEngine
    return OrbEngine(0, cap=4096)


@pytest.fixture(scope="module")
def eng8k():
    """Capacity for the 1080p preset (about 5k keypoints per frame)."""
    from visual_slam_amd.engine import OrbEngine
    return OrbEngine(0, cap=8192)


def test_golden_fixtures(gctx):
    golden_util.check_impl_against_golden(gctx)


@pytest.mark.parametrize("H,W,first,n,cell", [(96, 128, 0, 3, 12), (61, 83, 1000, 2, 12), (480, 640, 7, 2, 12),
                                              (1080, 1920, 3, 1, 20)])
def test_synth_frames_device_equals_numpy(eng, H, W, first, n, cell):
    g = eng.synth_frames(n, H, W, first_frame=first, cell=cell)
    eng.sync()
    host = g.cpu().numpy()
    for i in range(n):
        assert np.array_equal(host[i], synth.synth_frame(H, W, first + i, cell=cell))


def test_extract_batch_and_consecutive_matches(eng):
    import torch
    F, H, W = 5, 240, 320
    frames = np.stack([synth.synth_frame(H, W, f, cell=12) for f in range(F - 1)] + [make_test_image(H, W, 77)])
    gray = torch.from_numpy(frames).to(eng.device)
    feat = eng.extract_batch(gray)
    m = eng.match_consecutive(feat)
    eng.sync()
    ex = [oracle.c.extract_gray(frames[f]) for f in range(F)]
    counts = feat.counts.cpu().numpy()
    for f in range(F):
        assert counts[f].tolist() == [len(ex[f][0]), len(ex[f][1])]
        kp, desc, vidx = feat.frame(f)
        assert np.array_equal(bits(kp), bits(ex[f][0]))
        assert np.array_equal(desc, ex[f][1]) and np.array_equal(vidx, ex[f][2])
    for f in range(1, F):
        eidx, edist = oracle.c.knn2(ex[f][1], ex[f - 1][1])
        idx, dist, good = m.pair(f - 1, len(ex[f][1]))
        assert np.array_equal(idx, eidx) and np.array_equal(dist, edist)
        want = np.zeros(len(eidx), np.uint8)
        want[oracle.c.ratio_filter(eidx, edist)[0]] = 1
        assert np.array_equal(good, want)
    # explicit pair list, including a frame against itself and a reversed pair
    pairs = torch.tensor([[0, 0], [1, 3], [4, 2]], dtype=torch.int32, device=eng.device)
    mp = eng.match_pairs(feat, pairs)
    eng.sync()
    for p, (qf, tf) in enumerate(pairs.cpu().tolist()):
        eidx, edist = oracle.c.knn2(ex[qf][1], ex[tf][1])
        idx, dist, _ = mp.pair(p, len(ex[qf][1]))
        assert np.array_equal(idx, eidx) and np.array_equal(dist, edist)
    # pairs that name frames outside the batch are skipped, by both matcher families and by the ratio
    # filter; the pairs around them are computed as usual
    pairs = torch.tensor([[1, 0], [F, 0], [2, -1], [2**31 - 1, 1], [3, 2], [0, -2**31]], dtype=torch.int32,
                         device=eng.device)
    from visual_slam_amd import _lib
    for matcher in (_lib.MATCHER_MATRIX, _lib.MATCHER_VALU):
        eng.set_matcher(matcher)
        out = eng.alloc_matches(len(pairs), feat.cap)
        out.idx.fill_(-77); out.dist.fill_(-77); out.good.fill_(77)
        eng.match_pairs(feat, pairs, out=out)
        _, _, _, kept = eng.good_matches(feat, out, pairs)
        eng.sync()
        for p, (qf, tf) in enumerate(pairs.cpu().tolist()):
            if 0 <= qf < F and 0 <= tf < F:
                eidx, edist = oracle.c.knn2(ex[qf][1], ex[tf][1])
                idx, dist, _ = out.pair(p, len(ex[qf][1]))
                assert np.array_equal(idx, eidx) and np.array_equal(dist, edist)
                assert int(kept[p]) == len(oracle.c.ratio_filter(eidx, edist)[0])
            else:
                assert bool((out.idx[p] == -77).all()) and bool((out.dist[p] == -77).all())
                assert bool((out.good[p] == 77).all()) and int(kept[p]) == -1
    eng.set_matcher(_lib.MATCHER_MATRIX)


@pytest.mark.parametrize("frames,chunks", [(7, 3), (5, 5), (2, 4), (1, 4), (9, 2)])
def test_pipelined_extract_and_match_equals_separate_calls(eng, frames, chunks):
    """Two-stream pipeline over frame ranges: same features, same matches for every pair, including
    the pairs that join two ranges; run twice so buffers reused across calls are covered."""
    import torch
    H, W = 120, 168
    a = np.stack([synth.synth_frame(H, W, 40 + f, cell=12) for f in range(frames)])
    g = torch.from_numpy(a).to(eng.device)
    ref_f = eng.extract_batch(g)
    ref_m = eng.match_consecutive(ref_f)
    eng.sync()
    for _ in range(2):
        f = eng.alloc_features(frames)
        m = eng.alloc_matches(frames - 1)
        m.idx.fill_(-7); m.dist.fill_(-7); m.good.fill_(9)
        eng.extract_and_match(g, f, m, chunks=chunks)
        torch.cuda.synchronize()
        assert torch.equal(f.counts, ref_f.counts)
        for i in range(frames):
            n, nv = (int(v) for v in ref_f.counts[i].tolist())
            assert torch.equal(f.kp7[i, :n], ref_f.kp7[i, :n]) and torch.equal(f.desc[i, :nv], ref_f.desc[i, :nv])
            if i:
                assert torch.equal(m.idx[i - 1, :nv], ref_m.idx[i - 1, :nv])
                assert torch.equal(m.dist[i - 1, :nv], ref_m.dist[i - 1, :nv])
                assert torch.equal(m.good[i - 1, :nv], ref_m.good[i - 1, :nv])


def test_pipelined_matching_waits_for_its_own_extraction(eng8k):
    """Frames large enough that extraction takes longer than launching the matcher: the matching of a
    frame range must be ordered behind the extraction of that range on whatever stream the engine is
    bound to -- including the framework's default stream, whose handle is NULL."""
    eng = eng8k
    import torch
    P = synth.PRESETS["1080p"]
    frames, chunks = 20, 3
    assert eng.stream.cuda_stream == torch.cuda.current_stream(eng.device).cuda_stream
    for r in range(3):
        g = eng.synth_frames(frames, P["H"], P["W"], first_frame=3000 + 20 * r, cell=P["cell"], density=P["density"])
        ref_f = eng.extract_batch(g)
        ref_m = eng.match_consecutive(ref_f)
        torch.cuda.synchronize()
        f = eng.alloc_features(frames)
        m = eng.alloc_matches(frames - 1)
        m.idx.fill_(-7); m.dist.fill_(-7)
        eng.extract_and_match(g, f, m, chunks=chunks)
        torch.cuda.synchronize()
        assert torch.equal(f.counts, ref_f.counts)
        for i in range(1, frames):
            nv = int(ref_f.counts[i, 1])
            assert torch.equal(f.desc[i, :nv], ref_f.desc[i, :nv])
            assert torch.equal(m.idx[i - 1, :nv], ref_m.idx[i - 1, :nv]), (r, i)
            assert torch.equal(m.dist[i - 1, :nv], ref_m.dist[i - 1, :nv]), (r, i)


def test_engine_results_are_ordered_on_the_current_torch_stream(eng8k):
    """No explicit synchronisation: a torch operation queued on the current stream right after the
    engine calls must see their results (the library runs on the stream the engine was bound to)."""
    eng = eng8k
    import torch
    P = synth.PRESETS["1080p"]
    g = eng.synth_frames(12, P["H"], P["W"], first_frame=77, cell=P["cell"], density=P["density"])
    f = eng.alloc_features(12)
    f.counts.fill_(-1)
    eng.extract_batch(g, out=f)
    snap = f.counts.clone()                 # queued behind the extraction, no sync in between
    m = eng.match_consecutive(f)
    first = m.idx[:, :64].clone()
    torch.cuda.synchronize()
    assert (snap[:, 0] > 1000).all() and torch.equal(snap, f.counts)
    assert torch.equal(first, m.idx[:, :64]) and (first[:, :, 0] >= 0).all()


def test_two_frame_ranges_on_one_gpu_reproduce_the_single_batch(eng):
    """The seam logic of the multi-GPU scheduler with device tensors (everything but the transport):
    range B's first frame is matched against range A's last frame, which arrives as the packed
    boundary message; all pairs together equal the matches of the undivided sequence."""
    import torch
    from visual_slam_amd.scheduler import pack_boundary, unpack_boundary, partition_frames, boundary_message_bytes
    H, W, frames = 160, 224, 11
    a = np.stack([synth.synth_frame(H, W, 300 + f, cell=12) for f in range(frames)])
    g = torch.from_numpy(a).to(eng.device)
    whole_f = eng.extract_batch(g)
    whole_m = eng.match_consecutive(whole_f)
    (a0, a1), (b0, b1) = partition_frames(frames, 2)
    fa = eng.extract_batch(g[a0:a1])
    ma = eng.match_consecutive(fa)
    msg = pack_boundary(fa, fa.frames - 1, with_keypoints=True)
    assert msg.numel() == boundary_message_bytes(fa.cap, True) and msg.is_cuda
    wire = msg.clone()                                   # what the neighbour would receive
    fb = eng.extract_batch(g[b0:b1])
    mb = eng.match_consecutive(fb)
    bnd = unpack_boundary(wire, fb.cap, True)
    seam = eng.match_cross(fb, 0, bnd.desc, bnd.counts)
    torch.cuda.synchronize()
    assert bnd.counts[0].tolist() == whole_f.counts[a1 - 1].tolist()
    n_last = int(whole_f.counts[a1 - 1, 0])
    assert torch.equal(bnd.kp7[0, :n_last], whole_f.kp7[a1 - 1, :n_last])
    for p in range(frames - 1):                          # pair p = frame p + 1 against frame p
        nv = int(whole_f.counts[p + 1, 1])
        if p + 1 < a1:
            got = ma.idx[p, :nv], ma.dist[p, :nv]
        elif p + 1 == b0:
            got = seam.idx[0, :nv], seam.dist[0, :nv]
        else:
            got = mb.idx[p - b0, :nv], mb.dist[p - b0, :nv]
        assert torch.equal(got[0], whole_m.idx[p, :nv]) and torch.equal(got[1], whole_m.dist[p, :nv]), p


def test_batch_ratio_filter_lists_match_oracle(eng):
    """Device-side ratio-filter lists for consecutive pairs and for an explicit pair list."""
    import torch
    H, W = 160, 208
    a = np.stack([synth.synth_frame(H, W, 70 + f, cell=12) for f in range(4)])
    feat = eng.extract_batch(torch.from_numpy(a).to(eng.device))
    m = eng.match_consecutive(feat)
    gq, gt2, gt1, kept = eng.good_matches(feat, m)
    pairs = torch.tensor([[3, 0], [1, 1]], dtype=torch.int32, device=eng.device)
    mp = eng.match_pairs(feat, pairs)
    pq, pt2, pt1, pkept = eng.good_matches(feat, mp, pairs)
    eng.sync()
    descs = [oracle.c.extract_gray(a[i])[1] for i in range(4)]

    def check(qd, td, row_q, row_t2, row_t1, k):
        eidx, edist = oracle.c.knn2(qd, td)
        eq, e2, e1 = oracle.c.ratio_filter(eidx, edist)
        assert k == len(eq) and k > 0
        assert np.array_equal(row_q[:k], eq) and np.array_equal(row_t2[:k], e2) and np.array_equal(row_t1[:k], e1)

    for p in range(3):
        check(descs[p + 1], descs[p], gq[p].cpu().numpy(), gt2[p].cpu().numpy(), gt1[p].cpu().numpy(), int(kept[p]))
    check(descs[3], descs[0], pq[0].cpu().numpy(), pt2[0].cpu().numpy(), pt1[0].cpu().numpy(), int(pkept[0]))
    check(descs[1], descs[1], pq[1].cpu().numpy(), pt2[1].cpu().numpy(), pt1[1].cpu().numpy(), int(pkept[1]))
    # a pair whose train frame has a single descriptor: the reference's loop cannot unpack two neighbours
    desc = torch.zeros((2, feat.cap, 32), dtype=torch.uint8, device=eng.device)
    counts = torch.tensor([[1, 1], [5, 5]], dtype=torch.int32, device=eng.device)
    from visual_slam_amd.engine import BatchFeatures
    f2 = BatchFeatures(kp7=None, desc=desc, valid_idx=None, counts=counts, cap=feat.cap)
    m2 = eng.match_consecutive(f2)
    _, _, _, k2 = eng.good_matches(f2, m2)
    eng.sync()
    assert int(k2[0]) == -1


@pytest.mark.parametrize("bgr", [False, True])
def test_host_frame_stager_equals_whole_batch(eng, bgr):
    """Frames fed from host memory in chunks of 4 (10 frames: two full chunks and a short one): same
    features per frame and the same matches for every consecutive pair, also across chunk seams."""
    import torch
    from visual_slam_amd.staging import HostFrameStager
    H, W, F = 96, 144, 10
    gray = np.stack([synth.synth_frame(H, W, 300 + f, cell=12) for f in range(F)])
    if bgr:
        rng = np.random.default_rng(4)
        frames = rng.integers(0, 256, (F, H, W, 3), dtype=np.uint8)
        frames[..., 1] = gray                       # keep the blobs visible after the conversion
        gray_ref = np.stack([oracle.c.bgr2gray(f) for f in frames])
    else:
        frames, gray_ref = gray, gray
    ex = [oracle.c.extract_gray(g) for g in gray_ref]
    st = HostFrameStager(eng, H, W, chunk=4, bgr=bgr)
    seen = 0
    for ch in st.run(iter(frames)):
        torch.cuda.synchronize()
        assert ch.first == seen
        for i in range(ch.frames):
            f = ch.first + i
            kp, desc, vidx = ch.features.frame(i)
            assert np.array_equal(bits(kp), bits(ex[f][0])) and np.array_equal(desc, ex[f][1])
            assert np.array_equal(vidx, ex[f][2])
            idx, dist, good = ch.matches.pair(i, len(desc))
            if f == 0:
                assert (idx == -1).all()
            else:
                eidx, edist = oracle.c.knn2(ex[f][1], ex[f - 1][1])
                assert np.array_equal(idx, eidx) and np.array_equal(dist, edist)
        seen += ch.frames
    assert seen == F


@pytest.mark.parametrize("bgr", [False, True])
def test_frame_file_through_the_stager_equals_the_in_memory_path(eng, tmp_path, bgr):
    """A Y4M file on disk -> FrameFile -> pinned chunks -> device: the same features and matches as the
    frames handed over from memory (stand-in for cv.VideoCapture + cap.read(), reference main.py:80, views.py:81)."""
    import torch
    from visual_slam_amd import frames as fsrc
    from visual_slam_amd.staging import HostFrameStager
    H, W, F = 96, 144, 9
    gray = [synth.synth_frame(H, W, 500 + f, cell=12) for f in range(F)]
    path = str(tmp_path / "clip.y4m")
    fsrc.write_y4m(path, gray, colourspace="420jpeg")
    ex = [oracle.c.extract_gray(g) for g in gray]
    st = HostFrameStager(eng, H, W, chunk=4, bgr=bgr)
    seen = 0
    with fsrc.FrameFile(path, bgr=bgr) as src:
        for ch in st.run(src):
            torch.cuda.synchronize()
            for i in range(ch.frames):
                f = ch.first + i
                kp, desc, vidx = ch.features.frame(i)
                assert np.array_equal(bits(kp), bits(ex[f][0])) and np.array_equal(desc, ex[f][1])
                if f:
                    idx, dist, _ = ch.matches.pair(i, len(desc))
                    eidx, edist = oracle.c.knn2(ex[f][1], ex[f - 1][1])
                    assert np.array_equal(idx, eidx) and np.array_equal(dist, edist)
            seen += ch.frames
        assert seen == F and src.index == F


def test_parity_audit_of_a_batch_changes_nothing_and_reports_per_frame(eng):
    import torch
    H, W, F = 240, 320, 12
    gray = torch.from_numpy(np.stack([synth.synth_frame(H, W, 900 + f, cell=12) for f in range(F)])).to(eng.device)
    plain = eng.extract_batch(gray)
    eng.set_audit(True)
    try:
        audited = eng.extract_batch(gray)
        a = eng.audit_read(F)
    finally:
        eng.set_audit(False)
    eng.sync()
    assert a.shape == (F, 4) and a.min() >= 0 and int(a[:, 2].max()) == 0 and int(a[:, 3].max()) == 0
    # a handful of keypoints per frame: about 1.5e-8 expected per keypoint and per coordinate
    assert int(a[:, 0].sum()) <= 2 and int(a[:, 1].sum()) <= 8
    assert torch.equal(plain.counts, audited.counts)
    for f in range(F):
        for x, y in zip(plain.frame(f), audited.frame(f)):
            assert np.array_equal(np.ascontiguousarray(x).view(np.uint8), np.ascontiguousarray(y).view(np.uint8))


def test_extract_batch_capacity_overflow_keeps_first_rows(eng):
    import torch
    from visual_slam_amd.engine import OrbEngine
    H, W = 240, 320
    frames = np.stack([synth.synth_frame(H, W, f, cell=12) for f in range(2)])
    ex = [oracle.c.extract_gray(f) for f in frames]
    cap = 100
    assert all(len(e[0]) > cap for e in ex)
    small = OrbEngine(0, cap=cap)
    feat = small.extract_batch(torch.from_numpy(frames).to(small.device))
    m = small.match_consecutive(feat)
    small.sync()
    counts = feat.counts.cpu().numpy()
    assert feat.overflowed() == [0, 1]
    nvk = [int((e[2] < cap).sum()) for e in ex]                              # descriptors of the first cap keypoints
    assert min(nvk) < cap, "the scene must put border keypoints among the first rows"
    for f in range(2):
        # n is the true total; the second count is the number of descriptor rows actually stored
        assert counts[f].tolist() == [len(ex[f][0]), nvk[f]]
        kp = feat.kp7[f].cpu().numpy()
        assert np.array_equal(bits(kp), bits(ex[f][0][:cap]))                # first cap keypoints, in order
        desc = feat.desc[f, :nvk[f]].cpu().numpy()
        assert np.array_equal(desc, ex[f][1][:nvk[f]])
        assert np.array_equal(feat.valid_idx[f, :nvk[f]].cpu().numpy(), ex[f][2][:nvk[f]])
    # matching works on exactly the stored rows: equal to the oracle on the truncated descriptor sets
    eidx, edist = oracle.c.knn2(ex[1][1][:nvk[1]], ex[0][1][:nvk[0]])
    for matcher in (_lib.MATCHER_MATRIX, _lib.MATCHER_VALU):
        small.set_matcher(matcher)
        # poison the rows behind the stored ones: a matcher that read them would pick them up
        feat.desc[0, nvk[0]:] = torch.from_numpy(ex[1][1][:1]).to(small.device)
        m = small.match_consecutive(feat)
        small.sync()
        idx, dist, good = m.pair(0, nvk[1])
        assert np.array_equal(idx, eidx) and np.array_equal(dist, edist)
        gq, _, _ = oracle.c.ratio_filter(eidx, edist)
        assert np.array_equal(np.nonzero(good)[0], gq)
        _, _, _, kept = small.good_matches(feat, m)
        assert int(kept[0]) == len(gq)


def test_full_frame_noise_overflows_capacity_and_still_matches(eng8k):
    eng = eng8k
    """Uniform noise at 1080p: nearly every pixel neighbourhood is a corner candidate, far more than
    cap = 8192 per frame.  Extraction keeps the first cap keypoints, matching runs on their
    descriptors only, both equal to the CPU restatement cut at the same place."""
    import torch
    H, W = 1080, 1920
    rng = np.random.default_rng(77)
    frames = rng.integers(0, 256, size=(2, H, W), dtype=np.uint8)
    cap = eng.cap
    feat = eng.extract_batch(torch.from_numpy(frames).to(eng.device))
    m = eng.match_consecutive(feat)
    eng.sync()
    counts = feat.counts.cpu().numpy()
    ex = [oracle.c.extract_gray(f) for f in frames]
    assert all(len(e[0]) > cap for e in ex)
    nvk = [int((e[2] < cap).sum()) for e in ex]
    for f in range(2):
        assert counts[f].tolist() == [len(ex[f][0]), nvk[f]]
        kp, desc, vidx = feat.frame(f)
        assert np.array_equal(bits(kp), bits(ex[f][0][:cap]))
        assert np.array_equal(desc, ex[f][1][:nvk[f]]) and np.array_equal(vidx, ex[f][2][:nvk[f]])
    eidx, edist = oracle.c.knn2(ex[1][1][:nvk[1]], ex[0][1][:nvk[0]])
    idx, dist, _ = m.pair(0, nvk[1])
    assert np.array_equal(idx, eidx) and np.array_equal(dist, edist)


def test_cross_buffer_pair_with_device_counts(eng):
    import torch
    H, W = 200, 260
    a = np.stack([synth.synth_frame(H, W, f, cell=12) for f in (0, 1, 2)])
    b = np.stack([synth.synth_frame(H, W, f, cell=12) for f in (3, 4)])
    fa = eng.extract_batch(torch.from_numpy(a).to(eng.device))
    fb = eng.extract_batch(torch.from_numpy(b).to(eng.device))
    m = eng.match_cross(fb, 0, fa.desc, fa.counts, tframe=2)     # frame 3 queries frame 2
    eng.sync()
    q = oracle.c.extract_gray(b[0])[1]
    t = oracle.c.extract_gray(a[2])[1]
    eidx, edist = oracle.c.knn2(q, t)
    idx, dist, _ = m.pair(0, len(q))
    assert np.array_equal(idx, eidx) and np.array_equal(dist, edist)


def test_dense_matching_stress_20k(eng):
    """20k x 20k BRIEF-256 top-2 with 1 % duplicated rows (train-axis split + merge path)."""
    import torch
    q = synth.synth_descriptors(20000, 12345)
    t = synth.synth_descriptors(20000, 12346)
    t[123] = q[5]
    t[19999] = q[5]
    idx, dist, good = eng.match_sets(torch.from_numpy(q).to(eng.device), torch.from_numpy(t).to(eng.device))
    eng.sync()
    eidx, edist = oracle.c.knn2(q, t)
    assert np.array_equal(idx.cpu().numpy(), eidx) and np.array_equal(dist.cpu().numpy(), edist)
    assert idx[5].tolist() == [123, 19999] and dist[5].tolist() == [0, 0]
    want = np.zeros(len(q), np.uint8)
    want[oracle.c.ratio_filter(eidx, edist)[0]] = 1
    assert np.array_equal(good.cpu().numpy(), want)


def test_long_train_set_without_train_split(eng):
    """More than 8192 train rows handled by ONE workgroup column: with thousands of pairs in a launch
    the train axis is not split, so the matrix-core kernel has to close a run of float keys and
    start the next one.  The same pair is listed many times; every copy must equal the oracle."""
    import torch
    from visual_slam_amd import _lib
    cap, nt, nq, npairs = 9504, 9500, 37, 2100
    rng = np.random.default_rng(5)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    q[0] = t[8191]; t[8192] = t[8191]; t[9499] = t[8191]      # ties across the run boundary
    q[1] = t[0]; t[8200] = t[0]
    desc = torch.zeros((2, cap, 32), dtype=torch.uint8, device=eng.device)
    desc[0, :nt] = torch.from_numpy(t).to(eng.device)
    desc[1, :nq] = torch.from_numpy(q).to(eng.device)
    counts = torch.tensor([[nt, nt], [nq, nq]], dtype=torch.int32, device=eng.device)
    pairs = torch.tensor([[1, 0]] * npairs, dtype=torch.int32, device=eng.device)
    eidx, edist = oracle.c.knn2(q, t)
    assert eidx[0].tolist() == [8191, 8192] and eidx[1].tolist() == [0, 8200]
    from visual_slam_amd.engine import BatchFeatures
    feat = BatchFeatures(kp7=None, desc=desc, valid_idx=None, counts=counts, cap=cap)
    for m in (_lib.MATCHER_MATRIX, _lib.MATCHER_VALU):
        eng.ctx.set_matcher(m)
        try:
            out = eng.match_pairs(feat, pairs)
            eng.sync()
        finally:
            eng.ctx.set_matcher(_lib.MATCHER_MATRIX)
        idx = out.idx[:, :nq].cpu().numpy()
        dist = out.dist[:, :nq].cpu().numpy()
        assert (idx == eidx[None]).all() and (dist == edist[None]).all()
        del out


@pytest.mark.parametrize("nt", [5121, 5153, 7000, 8193, 12001, 17000])
def test_long_train_sets_in_one_workgroup_column(eng, nt):
    """Train sets of 160 to 530 tiles in one workgroup column (many pairs in the launch, so no
    train-axis split), around and beyond the 8192-row key runs.  Winners in the very last rows, ties
    between early and late rows, queries without any close row and low-entropy rows (many equal
    distances) must all come out as the CPU restatement has them."""
    import torch
    from visual_slam_amd.engine import BatchFeatures
    cap, nq, npairs = nt + 7, 96, 2100
    rng = np.random.default_rng(nt)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    t[nt // 2:: 97] = (rng.integers(0, 2, (len(t[nt // 2:: 97]), 32)) * 255).astype(np.uint8)   # low entropy
    q[0] = t[nt - 1]                               # exact hit in the last row only
    q[1] = t[3]; t[nt - 2] = t[3]                  # tie: first tiles and last tile
    late = min(5200, nt - 30)
    q[2] = t[late]; q[2, 0] ^= 1                   # near hit far into the set
    q[3] = t[nt - 1]; q[3, 5] ^= 0x81; t[10] = q[3]; t[10, 7] ^= 0x0F   # best late, second early
    q[4] = t[40]; t[41] = t[40]                    # both early
    q[5:9] = (rng.integers(0, 2, (4, 32)) * 255).astype(np.uint8)
    desc = torch.zeros((2, cap, 32), dtype=torch.uint8, device=eng.device)
    desc[0, :nt] = torch.from_numpy(t).to(eng.device)
    desc[1, :nq] = torch.from_numpy(q).to(eng.device)
    counts = torch.tensor([[nt, nt], [nq, nq]], dtype=torch.int32, device=eng.device)
    pairs = torch.tensor([[1, 0]] * npairs, dtype=torch.int32, device=eng.device)
    eidx, edist = oracle.c.knn2(q, t)
    assert eidx[0, 0] == nt - 1 and eidx[1].tolist() == [3, nt - 2] and eidx[4].tolist() == [40, 41]
    feat = BatchFeatures(kp7=None, desc=desc, valid_idx=None, counts=counts, cap=cap)
    out = eng.match_pairs(feat, pairs)
    eng.sync()
    idx = out.idx[:, :nq].cpu().numpy()
    dist = out.dist[:, :nq].cpu().numpy()
    assert (idx == eidx[None]).all() and (dist == edist[None]).all()


def test_matching_properties_self_match(eng):
    """Size-independent properties: a set matched against itself finds itself at distance 0 (lowest
    index among exact duplicates), dist0 <= dist1, indices in range; repeated runs are identical."""
    import torch
    d = synth.synth_descriptors(30000, 99, dup_frac=0.02)
    dd = torch.from_numpy(d).to(eng.device)
    idx, dist, _ = eng.match_sets(dd, dd)
    idx2, dist2, _ = eng.match_sets(dd, dd)
    eng.sync()
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    assert np.array_equal(idx, idx2.cpu().numpy()) and np.array_equal(dist, dist2.cpu().numpy())
    assert (dist[:, 0] == 0).all() and (dist[:, 0] <= dist[:, 1]).all()
    assert (idx >= 0).all() and (idx < len(d)).all() and (idx[:, 0] != idx[:, 1]).all()
    first_of = {}
    for i, row in enumerate(map(bytes, d)):
        first_of.setdefault(row, i)
    want0 = np.array([first_of[bytes(r)] for r in d])
    assert np.array_equal(idx[:, 0], want0)


def test_full_size_1080p_batch_properties(eng):
    """1080p batch: keypoints ascending in (y, x), descriptor rows in keypoint order and only for
    keypoints >= 16 px from every edge, counts consistent; one frame checked against the oracle."""
    import torch
    from visual_slam_amd.engine import OrbEngine
    P = synth.PRESETS["1080p"]
    big = OrbEngine(0, cap=8192)
    F = 12
    gray = big.synth_frames(F, P["H"], P["W"], first_frame=40, cell=P["cell"], density=P["density"])
    feat = big.extract_batch(gray)
    feat2 = big.extract_batch(gray)
    big.sync()
    counts = feat.counts.cpu().numpy()
    assert np.array_equal(counts, feat2.counts.cpu().numpy())
    for f in range(F):
        kp, desc, vidx = feat.frame(f)
        n, nv = counts[f]
        assert 4000 < n < 6000 and len(kp) == n and len(desc) == nv
        x, y = kp[:, 0].astype(np.int64), kp[:, 1].astype(np.int64)
        key = y * 100000 + x
        assert (np.diff(key) > 0).all()
        assert (x >= 3).all() and (x < P["W"] - 3).all() and (y >= 3).all() and (y < P["H"] - 3).all()
        inner = (x >= 16) & (x <= P["W"] - 17) & (y >= 16) & (y <= P["H"] - 17)
        assert np.array_equal(vidx, np.nonzero(inner)[0])
        assert (kp[:, 2] == 7).all() and (kp[:, 4] == 0).all() and (kp[:, 5] == 0).all() and (kp[:, 6] == -1).all()
        assert (kp[:, 3] >= 0).all() and (kp[:, 3] <= 360).all()
        k2, d2, v2 = feat2.frame(f)
        assert np.array_equal(bits(kp), bits(k2)) and np.array_equal(desc, d2)
    host = gray[5].cpu().numpy()
    ekp, edesc, evidx = oracle.c.extract_gray(host)
    kp, desc, vidx = feat.frame(5)
    assert np.array_equal(bits(kp), bits(ekp)) and np.array_equal(desc, edesc) and np.array_equal(vidx, evidx)


def test_4k_frames_pipelined_against_oracle():
    """3840x2160 frames (~20k keypoints each, capacity 32768) through the pipelined batch path: every
    frame's features and every consecutive pair's top-2 equal the CPU restatement."""
    import torch
    from visual_slam_amd.engine import OrbEngine
    P = synth.PRESETS["4k"]
    e = OrbEngine(0, cap=32768)
    F = 3
    gray = e.synth_frames(F, P["H"], P["W"], first_frame=7, cell=P["cell"], density=P["density"])
    feat = e.alloc_features(F)
    m = e.alloc_matches(F - 1)
    e.extract_and_match(gray, feat, m, chunks=2)
    torch.cuda.synchronize()
    host = gray.cpu().numpy()
    ex = [oracle.c.extract_gray(host[i]) for i in range(F)]
    for i in range(F):
        kp, desc, vidx = feat.frame(i)
        assert 15000 < len(kp) < 25000
        assert np.array_equal(bits(kp), bits(ex[i][0])) and np.array_equal(desc, ex[i][1])
        assert np.array_equal(vidx, ex[i][2])
        if i:
            eidx, edist = oracle.c.knn2(ex[i][1], ex[i - 1][1])
            idx, dist, good = m.pair(i - 1, len(desc))
            assert np.array_equal(idx, eidx) and np.array_equal(dist, edist)
            want = np.zeros(len(desc), np.uint8)
            want[oracle.c.ratio_filter(eidx, edist)[0]] = 1
            assert np.array_equal(good, want)


def test_device_libm_parity_gate_orientation(gctx):
    """Orientation goes through double atan2 on the device and through glibc in the reference; the
    float angle must agree bit for bit.  ~2.8 M keypoints on images with strong gradients."""
    mism = 0
    total = 0
    for seed in (1, 2):
        rng = np.random.default_rng(seed)
        H, W = 1000, 1400
        yy, xx = np.mgrid[0:H, 0:W]
        img = (127 + 100 * np.sin(xx / (7.0 + seed)) * np.cos(yy / 11.0) + rng.integers(-25, 26, (H, W)))
        img = np.clip(img, 0, 255).astype(np.uint8)
        xy = np.stack([xx.reshape(-1), yy.reshape(-1)], axis=1).astype(np.int32)
        ang = gctx.orient(img, xy)
        exp = oracle.c.orient(img, xy)
        mism += int((bits(ang) != bits(exp)).sum())
        total += len(xy)
    assert total >= 2_800_000
    assert mism == 0, "%d of %d float angles differ from the host libm result" % (mism, total)


def test_device_libm_parity_gate_descriptors(gctx):
    """Steered BRIEF goes through double cos/sin on the device: 200k keypoints at random float
    angles (plus the exact axes) on a structured 1080p image must give identical descriptors."""
    img = make_test_image(1080, 1920, 31, nrect=4000)
    rng = np.random.default_rng(32)
    n = 200_000
    xy = np.stack([rng.integers(0, 1920, n), rng.integers(0, 1080, n)], axis=1).astype(np.int32)
    ang = rng.uniform(0, 360, n).astype(np.float32)
    ang[:5] = [0, 90, 180, 270, 360]
    ang[5:1005] = np.round(ang[5:1005])          # whole degrees
    desc, vidx = gctx.describe(img, xy, ang)
    edesc, evidx = oracle.c.describe(img, xy, ang)
    assert np.array_equal(vidx, evidx)
    bad = int((desc != edesc).any(axis=1).sum())
    assert bad == 0, "%d of %d descriptors differ" % (bad, len(edesc))


def test_large_launch_uses_64_keypoints_per_wave_and_matches_oracle():
    """A batch big enough (frames x capacity >= 2^19) to take the 64-keypoints-per-wave variant of the
    orientation / descriptor kernel; frames spot-checked against the oracle, all frames for counts."""
    import torch
    from visual_slam_amd.engine import OrbEngine
    P = synth.PRESETS["480p"]
    e = OrbEngine(0, cap=8192)
    F = 64
    gray = e.synth_frames(F, P["H"], P["W"], first_frame=500, cell=P["cell"], density=P["density"])
    feat = e.extract_batch(gray)
    m = e.match_consecutive(feat)
    e.sync()
    host = gray.cpu().numpy()
    counts = feat.counts.cpu().numpy()
    prev = None
    for f in (0, 1, 31, 62, 63):
        ekp, edesc, evidx = oracle.c.extract_gray(host[f])
        kp, desc, vidx = feat.frame(f)
        assert counts[f].tolist() == [len(ekp), len(edesc)]
        assert np.array_equal(bits(kp), bits(ekp)) and np.array_equal(desc, edesc) and np.array_equal(vidx, evidx)
        if prev is not None and prev[0] == f - 1:
            eidx, edist = oracle.c.knn2(edesc, prev[1])
            idx, dist, _ = m.pair(f - 1, len(edesc))
            assert np.array_equal(idx, eidx) and np.array_equal(dist, edist)
        prev = (f, edesc)


def test_describe_with_non_finite_and_huge_angles(gctx):
    """Caller-supplied angles that are NaN / infinite make every rotated coordinate NaN; the
    reference's cvRound then returns INT_MIN, every sample is outside the image and reads 0, so the
    descriptor is all zeros.  Huge finite angles go through the library's argument reduction."""
    img = make_test_image(100, 140, 41)
    xy = np.array([[50, 50], [60, 40], [70, 55], [40, 60], [90, 50], [30, 30]], np.int32)
    ang = np.array([np.nan, np.inf, -np.inf, 1e30, -12345.678, 3.0e5], np.float32)
    desc, vidx = gctx.describe(img, xy, ang)
    edesc, evidx = oracle.c.describe(img, xy, ang)
    assert np.array_equal(vidx, evidx) and np.array_equal(desc, edesc)
    assert not desc[:3].any()


def test_reserve_presizes_the_work_buffers(eng):
    """After orbamd_ctx_reserve a batch call of that size finds its buffers in place; results are the
    same as without it, and nonsense sizes are rejected."""
    import torch
    from visual_slam_amd.engine import OrbEngine
    H, W = 120, 160
    frames = np.stack([synth.synth_frame(H, W, f, cell=12) for f in range(3)])
    e2 = OrbEngine(0, cap=512)
    e2.ctx.reserve(3, H, W, 512)
    f2 = e2.extract_batch(torch.from_numpy(frames).to(e2.device))
    m2 = e2.match_consecutive(f2)
    e2.sync()
    for f in range(3):
        ekp, edesc, evidx = oracle.c.extract_gray(frames[f])
        kp, desc, vidx = f2.frame(f)
        assert np.array_equal(bits(kp), bits(ekp)) and np.array_equal(desc, edesc) and np.array_equal(vidx, evidx)
    with pytest.raises(_lib.OrbAmdError):
        e2.ctx.reserve(0, H, W, 512)
    with pytest.raises(_lib.OrbAmdError):
        e2.ctx.reserve(3, H, W, 0)


@pytest.mark.parametrize("scene,H,W,frames", [("sparse", 1080, 1920, 3), ("textured", 1080, 1920, 2),
                                              ("sparse", 50, 48, 3), ("textured", 64, 240, 2),
                                              ("textured", 65, 256, 2), ("noise", 130, 496, 2),
                                              ("noise", 64, 960, 3), ("noise", 200, 1920, 1)])
def test_tile_and_per_keypoint_describe_kernels_agree(eng8k, scene, H, W, frames):
    """The frame-batch path describes keypoints tile by tile when the frames are made of whole aligned
    16-byte units, and keypoint by keypoint otherwise.  The same frames at a 16-byte-aligned address and
    at an address 4 bytes further on take one path each; every output must be identical (bit patterns of
    the angles included)."""
    import torch
    eng = eng8k
    if scene == "sparse":
        g = eng.synth_frames(frames, H, W, first_frame=11, cell=12)
    elif scene == "textured":
        g = eng.synth_textured_frames(frames, H, W, first_frame=5)
    else:
        g = torch.from_numpy(np.random.default_rng(3).integers(0, 256, (frames, H, W), dtype=np.uint8)).to(eng.device)
    assert g.data_ptr() % 16 == 0 and (H * W) % 16 == 0
    raw = torch.empty(frames * H * W + 16, dtype=torch.uint8, device=eng.device)
    shifted = raw[4:4 + frames * H * W].view(frames, H, W)
    shifted.copy_(g)
    assert shifted.data_ptr() % 16 == 4
    a = eng.extract_batch(g)
    b = eng.extract_batch(shifted)
    torch.cuda.synchronize()
    assert torch.equal(a.counts, b.counts) and int(a.counts[:, 1].min()) > 0
    for i in range(frames):
        n, nv = (min(int(v), a.cap) for v in a.counts[i])
        assert torch.equal(a.kp7[i, :n].view(torch.int32), b.kp7[i, :n].view(torch.int32))
        assert torch.equal(a.desc[i, :nv], b.desc[i, :nv])
        assert torch.equal(a.valid_idx[i, :nv], b.valid_idx[i, :nv])
