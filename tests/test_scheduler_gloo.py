"""Frame-range scheduler over torch.distributed (gloo, world_size 2, CPU): partitioning, the
fixed-size boundary message, and the seam pair.  The compute engine is the CPU restatement."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from visual_slam_amd import synth  # noqa: E402
from visual_slam_amd.scheduler import (FrameRangeScheduler, boundary_message_bytes, pack_boundary,  # noqa: E402
                                        partition_frames, unpack_boundary)

H, W, CELL, TOTAL, CAP = 96, 128, 12, 7, 256


def test_partition_frames():
    assert partition_frames(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert partition_frames(8, 8) == [(i, i + 1) for i in range(8)]
    assert partition_frames(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]
    assert partition_frames(0, 2) == [(0, 0), (0, 0)]
    for total in (1, 17, 1000, 4001):
        for world in (1, 2, 4, 8):
            parts = partition_frames(total, world)
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [e - s for s, e in parts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        partition_frames(5, 0)


def test_boundary_message_roundtrip():
    from oracle_engine import OracleEngine
    eng = OracleEngine(cap=CAP)
    gray = torch.from_numpy(synth.synth_frames(H, W, 0, 2, cell=CELL))
    feat = eng.extract_batch(gray)
    for with_kp in (True, False):
        msg = pack_boundary(feat, 1, with_kp)
        assert msg.numel() == boundary_message_bytes(CAP, with_kp) and msg.dtype == torch.uint8
        b = unpack_boundary(msg.clone(), CAP, with_kp)
        assert torch.equal(b.counts[0], feat.counts[1])
        assert torch.equal(b.desc[0], feat.desc[1])
        if with_kp:
            assert torch.equal(b.kp7[0].view(torch.int32), feat.kp7[1].view(torch.int32))
        else:
            assert b.kp7 is None


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir, total=TOTAL):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle_engine import OracleEngine
        s, e = partition_frames(total, world)[rank]
        gray = (torch.from_numpy(synth.synth_frames(H, W, s, e - s, cell=CELL)) if e > s
                else torch.zeros((0, H, W), dtype=torch.uint8))
        sched = FrameRangeScheduler(OracleEngine(cap=CAP))
        assert (sched.rank, sched.world) == (rank, world)
        res = sched.process(gray, first_frame=s)
        if res.features is None:                     # a rank without frames only relays the seam frame
            assert e == s
            np.savez(os.path.join(outdir, "rank%d.npz" % rank), first=s, empty=1)
            dist.barrier()
            return
        out = {"first": s, "counts": res.features.counts.numpy(), "sent": res.wire_bytes_sent,
               "received": res.wire_bytes_received,
               "idx": res.matches.idx.numpy(), "dist": res.matches.dist.numpy(), "good": res.matches.good.numpy()}
        if res.boundary_matches is not None:
            out.update(b_idx=res.boundary_matches.idx.numpy(), b_dist=res.boundary_matches.dist.numpy(),
                       b_good=res.boundary_matches.good.numpy(), b_counts=res.boundary.counts.numpy(),
                       b_kp7=res.boundary.kp7.numpy())
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_ranks_reproduce_the_single_process_matches(tmp_path):
    import oracle
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    frames = synth.synth_frames(H, W, 0, TOTAL, cell=CELL)
    ex = [oracle.c.extract_gray(f) for f in frames]
    assert min(len(e[1]) for e in ex) > 20
    parts = partition_frames(TOTAL, world)
    seen_pairs = set()
    for rank, (s, e) in enumerate(parts):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        assert int(z["first"]) == s
        for f in range(s, e):
            assert z["counts"][f - s].tolist() == [len(ex[f][0]), len(ex[f][1])]
        for f in range(s + 1, e):                     # local pairs
            nq = len(ex[f][1])
            eidx, edist = oracle.c.knn2(ex[f][1], ex[f - 1][1])
            assert np.array_equal(z["idx"][f - s - 1, :nq], eidx) and np.array_equal(z["dist"][f - s - 1, :nq], edist)
            want_good = np.zeros(nq, np.uint8)
            want_good[oracle.c.ratio_filter(eidx, edist)[0]] = 1
            assert np.array_equal(z["good"][f - s - 1, :nq], want_good)
            seen_pairs.add(f)
        if rank > 0:                                   # the seam pair (s, s - 1)
            nq = len(ex[s][1])
            eidx, edist = oracle.c.knn2(ex[s][1], ex[s - 1][1])
            assert np.array_equal(z["b_idx"][0, :nq], eidx) and np.array_equal(z["b_dist"][0, :nq], edist)
            assert z["b_counts"][0].tolist() == [len(ex[s - 1][0]), len(ex[s - 1][1])]
            n_prev = len(ex[s - 1][0])
            assert np.array_equal(z["b_kp7"][0, :n_prev].view(np.uint32), ex[s - 1][0].view(np.uint32))
            seen_pairs.add(s)
            # only the live rows travelled: 16-byte header + nd descriptors + n keypoint rows
            assert int(z["received"]) == 16 + 32 * len(ex[s - 1][1]) + 28 * n_prev
        else:
            assert "b_idx" not in z.files and int(z["received"]) == 0
        if rank + 1 < world:
            assert int(z["sent"]) == 16 + 32 * len(ex[e - 1][1]) + 28 * len(ex[e - 1][0])
    assert seen_pairs == set(range(1, TOTAL))          # every consecutive pair exactly once


@pytest.mark.timeout(180)
def test_more_ranks_than_frames_does_not_deadlock(tmp_path):
    """3 ranks, 2 frames: ranges (0,1), (1,2), (2,2).  The rank without frames must still take part in
    the seam chain, or its neighbour blocks in its send."""
    import oracle
    world, total = 3, 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), total), nprocs=world, join=True)
    frames = synth.synth_frames(H, W, 0, total, cell=CELL)
    ex = [oracle.c.extract_gray(f) for f in frames]
    z1 = np.load(os.path.join(str(tmp_path), "rank1.npz"))
    nq = len(ex[1][1])
    eidx, edist = oracle.c.knn2(ex[1][1], ex[0][1])
    assert np.array_equal(z1["b_idx"][0, :nq], eidx) and np.array_equal(z1["b_dist"][0, :nq], edist)
    assert "empty" in np.load(os.path.join(str(tmp_path), "rank2.npz")).files
