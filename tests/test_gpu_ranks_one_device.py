"""GPU, one device: the frame-range scheduler with several ranks that SHARE one GPU.  RCCL refuses to put
two ranks on one device, so the process group is gloo and the boundary frame is staged through host memory;
everything else is the multi-GPU path as it runs under RCCL: one process per rank started by
torch.distributed.run, the boundary frame extracted first, its counts read through pinned memory, the
transfers on a side stream, the seam pair matched by orbamd_knn2_cross_dev."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle
from visual_slam_amd import synth
from visual_slam_amd.scheduler import partition_frames

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
H, W, CELL, CAP = 240, 320, 12, 2048


def _run(tmp_path, world, total):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(HERE, "nccl_seam_worker.py"),
           str(tmp_path), str(H), str(W), str(CELL), str(total), str(CAP), "gloo"]
    subprocess.run(cmd, env=env, check=True, timeout=240)


def _check(tmp_path, world, total):
    frames = synth.synth_frames(H, W, 0, total, cell=CELL)
    ex = [oracle.c.extract_gray(f) for f in frames]
    assert min(len(e[1]) for e in ex) > 20 and max(len(e[0]) for e in ex) <= CAP
    seen = set()
    prev_last = None          # last frame that exists before the current rank's range
    for rank, (s, e) in enumerate(partition_frames(total, world)):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        assert int(z["first"]) == s
        if e == s:
            assert "empty" in z.files
            continue
        for f in range(s, e):
            assert z["counts"][f - s].tolist() == [len(ex[f][0]), len(ex[f][1])]
        for f in range(s + 1, e):
            nq = len(ex[f][1])
            eidx, edist = oracle.c.knn2(ex[f][1], ex[f - 1][1])
            assert np.array_equal(z["idx"][f - s - 1, :nq], eidx) and np.array_equal(z["dist"][f - s - 1, :nq], edist)
            want = np.zeros(nq, np.uint8)
            want[oracle.c.ratio_filter(eidx, edist)[0]] = 1
            assert np.array_equal(z["good"][f - s - 1, :nq], want)
            seen.add(f)
        if rank > 0 and prev_last is not None:
            nq = len(ex[s][1])
            eidx, edist = oracle.c.knn2(ex[s][1], ex[prev_last][1])
            assert np.array_equal(z["b_idx"][0, :nq], eidx) and np.array_equal(z["b_dist"][0, :nq], edist)
            assert z["b_counts"][0].tolist() == [len(ex[prev_last][0]), len(ex[prev_last][1])]
            n_prev = len(ex[prev_last][0])
            assert np.array_equal(z["b_kp7"][0, :n_prev].view(np.uint32), ex[prev_last][0].view(np.uint32))
            assert int(z["received"]) == 16 + 32 * len(ex[prev_last][1]) + 28 * n_prev
            seen.add(s)
        prev_last = e - 1
    assert seen == set(range(1, total))


@pytest.mark.timeout(300)
def test_two_ranks_on_one_gpu_reproduce_the_single_process_matches(tmp_path):
    _run(tmp_path, 2, 7)
    _check(tmp_path, 2, 7)


@pytest.mark.timeout(300)
def test_four_ranks_one_of_them_without_frames(tmp_path):
    _run(tmp_path, 4, 3)         # ranges (0,1) (1,2) (2,3) (3,3): single-frame ranks and an empty one
    _check(tmp_path, 4, 3)
