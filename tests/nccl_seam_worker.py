"""One rank of the multi-rank seam tests (started under torch.distributed.run by tests/test_gpu_nccl.py and
tests/test_gpu_ranks_one_device.py): extracts and matches its contiguous frame range on its GPU through
FrameRangeScheduler.process and writes what it got to <outdir>/rank<r>.npz.  With backend "nccl" (default)
every rank has its own GPU and the boundary frame travels over RCCL send/recv; with "gloo" the ranks share
the visible GPUs round-robin and the boundary frame is staged through host memory."""
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
    backend = sys.argv[7] if len(sys.argv) > 7 else "nccl"
    rank, local, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
        dist.init_process_group(backend)
    try:
        s, e = partition_frames(total, world)[rank]
        eng = OrbEngine(local, cap=cap)
        gray = (torch.from_numpy(synth.synth_frames(H, W, s, e - s, cell=cell)).to(eng.device) if e > s
                else torch.zeros((0, H, W), dtype=torch.uint8, device=eng.device))
        res = FrameRangeScheduler(eng).process(gray, first_frame=s)
        torch.cuda.synchronize()
        if res.features is None:
            np.savez(os.path.join(outdir, "rank%d.npz" % rank), first=s, empty=1)
            dist.barrier()
            return
        out = {"first": s, "world": dist.get_world_size(), "device": torch.cuda.current_device(),
               "sent": res.wire_bytes_sent, "received": res.wire_bytes_received,
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
