"""GPU, two devices: the frame-range scheduler over RCCL (torch.distributed backend "nccl"), one
process per GPU.  Mirror of tests/test_scheduler_gloo.py with the HIP library as the engine and
device tensors in the boundary message.  Skipped on a one-GPU machine."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle
from visual_slam_amd import synth
from visual_slam_amd.scheduler import partition_frames

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
H, W, CELL, TOTAL, CAP = 240, 320, 12, 7, 2048


@pytest.mark.timeout(300)
def test_two_gpu_ranks_reproduce_the_single_process_matches(tmp_path):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL does not run two ranks on one device)")
    world = 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(HERE, "nccl_seam_worker.py"),
           str(tmp_path), str(H), str(W), str(CELL), str(TOTAL), str(CAP)]
    subprocess.run(cmd, env=env, check=True, timeout=240)

    frames = synth.synth_frames(H, W, 0, TOTAL, cell=CELL)
    ex = [oracle.c.extract_gray(f) for f in frames]
    assert min(len(e[1]) for e in ex) > 20 and max(len(e[0]) for e in ex) <= CAP
    seen = set()
    devices = set()
    for rank, (s, e) in enumerate(partition_frames(TOTAL, world)):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        assert int(z["first"]) == s and int(z["world"]) == world
        devices.add(int(z["device"]))
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
        if rank > 0:
            nq = len(ex[s][1])
            eidx, edist = oracle.c.knn2(ex[s][1], ex[s - 1][1])
            assert np.array_equal(z["b_idx"][0, :nq], eidx) and np.array_equal(z["b_dist"][0, :nq], edist)
            assert z["b_counts"][0].tolist() == [len(ex[s - 1][0]), len(ex[s - 1][1])]
            n_prev = len(ex[s - 1][0])
            assert np.array_equal(z["b_kp7"][0, :n_prev].view(np.uint32), ex[s - 1][0].view(np.uint32))
            seen.add(s)
    assert seen == set(range(1, TOTAL)) and len(devices) == world
