"""One rank of the two-GPU seam test (started by tests/test_gpu_nccl.py under torch.distributed.run):
extracts and matches its contiguous frame range on its own GPU through FrameRangeScheduler.process,
with the boundary frame travelling over RCCL send/recv, and writes what it got to <outdir>/rank<r>.npz."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from visual_slam_amd import synth  # noqa: E402
from visual_slam_amd.engine import OrbEngine  # noqa: E402
from visual_slam_amd.scheduler import FrameRangeScheduler, partition_frames  # noqa: E402


def main():
    outdir, H, W, cell, total, cap = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), \
        int(sys.argv[5]), int(sys.argv[6])
    rank, local, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    try:
        s, e = partition_frames(total, world)[rank]
        eng = OrbEngine(local, cap=cap)
        gray = torch.from_numpy(synth.synth_frames(H, W, s, e - s, cell=cell)).to(eng.device)
        res = FrameRangeScheduler(eng).process(gray, first_frame=s)
        torch.cuda.synchronize()
        out = {"first": s, "world": dist.get_world_size(), "device": torch.cuda.current_device(),
               "counts": res.features.counts.cpu().numpy(), "idx": res.matches.idx.cpu().numpy(),
               "dist": res.matches.dist.cpu().numpy(), "good": res.matches.good.cpu().numpy()}
        if res.boundary_matches is not None:
            out.update(b_idx=res.boundary_matches.idx.cpu().numpy(), b_dist=res.boundary_matches.dist.cpu().numpy(),
                       b_good=res.boundary_matches.good.cpu().numpy(), b_counts=res.boundary.counts.cpu().numpy(),
                       b_kp7=res.boundary.kp7.cpu().numpy())
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
