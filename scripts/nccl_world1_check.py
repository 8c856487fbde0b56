"""World-size-1 RCCL check on one GPU: the process group, a barrier and an all-reduce work in the
same process as liborbamd.so (one HIP runtime), in the order bench.py uses them.
    RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 python scripts/nccl_world1_check.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from visual_slam_amd.engine import OrbEngine
from visual_slam_amd.scheduler import FrameRangeScheduler, pack_boundary, unpack_boundary

torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
eng = OrbEngine(0, cap=4096)
gray = eng.synth_frames(8, 480, 640, cell=12)
feat = eng.alloc_features(8)
matches = eng.alloc_matches(7)
eng.extract_and_match(gray, feat, matches, chunks=2)
sched = FrameRangeScheduler(eng, 0, 1)
reqs, rbuf = sched.exchange(feat)
assert reqs == [] and rbuf is None
msg = pack_boundary(feat, 7, True)                      # what rank r would send to rank r + 1
b = unpack_boundary(msg, feat.cap, True)
seam = eng.match_cross(feat, 0, b.desc, b.counts)
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
print("ok", feat.counts[:2].tolist(), float(t.item()), int((seam.idx[0, :10, 0] >= 0).sum()))
dist.destroy_process_group()
