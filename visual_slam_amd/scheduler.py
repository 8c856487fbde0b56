"""Frame-range scheduler: contiguous frame ranges per GPU, neighbour exchange of the boundary frame.

The reference processes one video on one thread, matching every frame against the previous one
(reference views.py:75-90).  Extraction is independent per frame and matching couples only frames
f and f-1, so a frame sequence shards into contiguous ranges: rank r owns frames [s_r, e_r), and
the only pair it cannot form by itself is (s_r, s_r - 1), whose train side lives on rank r-1.
Rank r-1 therefore sends the features of its LAST frame to rank r (one point-to-point message to
its direct neighbour; RCCL send/recv when the tensors are on GPUs) and nothing else crosses ranks.

The message has a fixed size (capacity-padded descriptors, optionally keypoint rows, plus the two
counts), so neither side needs a host round trip to learn a length before posting the transfer.

One process per GPU; torch.distributed provides the transport ("nccl" = RCCL for GPU tensors,
"gloo" for CPU tensors in tests).
"""
from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist

HEADER_BYTES = 16  # int32 n, nv, cap, has_keypoints


def partition_frames(total, world):
    """Contiguous ranges [start, end) per rank, sizes balanced to +-1, earlier ranks get the extra."""
    if total < 0 or world <= 0:
        raise ValueError("total >= 0 and world > 0 required")
    base, extra = divmod(total, world)
    out, s = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((s, s + n))
        s += n
    return out


def boundary_message_bytes(cap, with_keypoints=True):
    return HEADER_BYTES + cap * 32 + (cap * 28 if with_keypoints else 0)


def pack_boundary(feat, frame, with_keypoints=True):
    """Features of one frame -> one contiguous uint8 message on the same device."""
    cap = feat.cap
    msg = torch.empty(boundary_message_bytes(cap, with_keypoints), dtype=torch.uint8, device=feat.desc.device)
    hdr = msg[:HEADER_BYTES].view(torch.int32)
    hdr[0:2] = feat.counts[frame]
    hdr[2] = cap
    hdr[3] = 1 if with_keypoints else 0
    msg[HEADER_BYTES:HEADER_BYTES + cap * 32] = feat.desc[frame].reshape(-1)
    if with_keypoints:
        msg[HEADER_BYTES + cap * 32:] = feat.kp7[frame].reshape(-1).view(torch.uint8)
    return msg


@dataclass
class Boundary:
    """A received boundary frame, as views into the message buffer (no copies)."""
    counts: torch.Tensor           # int32 [1, 2]
    desc: torch.Tensor             # uint8 [1, cap, 32]
    kp7: Optional[torch.Tensor]    # float32 [1, cap, 7] or None


def unpack_boundary(msg, cap, with_keypoints=True):
    hdr = msg[:HEADER_BYTES].view(torch.int32)
    counts = hdr[0:2].view(1, 2)
    desc = msg[HEADER_BYTES:HEADER_BYTES + cap * 32].view(1, cap, 32)
    kp7 = None
    if with_keypoints:
        kp7 = msg[HEADER_BYTES + cap * 32:].view(torch.float32).view(1, cap, 7)
    return Boundary(counts=counts, desc=desc, kp7=kp7)


@dataclass
class ShardResult:
    features: object                 # BatchFeatures of this rank's frames
    matches: object                  # BatchMatches for local pairs (f, f-1), f = 1 .. frames-1
    boundary_matches: object = None  # BatchMatches (1 pair): local frame 0 vs the previous rank's last frame
    boundary: Optional[Boundary] = None
    first_frame: int = 0


class FrameRangeScheduler:
    """Runs one rank's share of a frame sequence and stitches the seam to the previous rank.

    `engine` provides extract_batch(gray), match_consecutive(features) and
    match_cross(features, qframe, t_desc, t_counts) (visual_slam_amd.engine.OrbEngine).
    """

    def __init__(self, engine, rank=None, world=None, group=None, with_keypoints=True):
        self.engine = engine
        self.group = group
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.rank, self.world = rank, world
        self.with_keypoints = with_keypoints

    def exchange(self, feat):
        """Send this rank's last frame to rank+1, receive rank-1's last frame.  Returns
        (requests, receive buffer or None); the transfer is in flight until wait()."""
        if self.world == 1:
            return [], None
        ops, rbuf = [], None
        cap = feat.cap
        if self.rank + 1 < self.world:
            sbuf = pack_boundary(feat, feat.frames - 1, self.with_keypoints)
            ops.append(dist.P2POp(dist.isend, sbuf, self.rank + 1, self.group))
        if self.rank > 0:
            rbuf = torch.empty(boundary_message_bytes(cap, self.with_keypoints), dtype=torch.uint8,
                               device=feat.desc.device)
            ops.append(dist.P2POp(dist.irecv, rbuf, self.rank - 1, self.group))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        return reqs, rbuf

    def _relay(self, device):
        """A rank without frames keeps the seam chain intact: it passes the previous rank's boundary
        frame on to the next rank unchanged (rank 0 without frames sends an empty frame), so its
        neighbours' transfers complete and the next rank that has frames still matches against the
        last frame that exists."""
        nbytes = boundary_message_bytes(self.engine.cap, self.with_keypoints)
        if self.rank > 0:
            buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
            for r in dist.batch_isend_irecv([dist.P2POp(dist.irecv, buf, self.rank - 1, self.group)]):
                r.wait()
        else:
            buf = torch.zeros(nbytes, dtype=torch.uint8, device=device)
            hdr = buf[:HEADER_BYTES].view(torch.int32)
            hdr[2] = self.engine.cap
            hdr[3] = 1 if self.with_keypoints else 0
        if self.rank + 1 < self.world:
            for r in dist.batch_isend_irecv([dist.P2POp(dist.isend, buf, self.rank + 1, self.group)]):
                r.wait()
        return buf

    def process(self, gray, first_frame=0):
        """gray: this rank's frames [frames, H, W].  Returns a ShardResult; all work is enqueued
        asynchronously where the engine is asynchronous.  A rank whose range is empty (more ranks
        than frames) returns a ShardResult without features."""
        eng = self.engine
        if gray.shape[0] == 0:
            if self.world > 1:
                self._relay(gray.device)
            return ShardResult(features=None, matches=None, first_frame=first_frame)
        feat = eng.extract_batch(gray)
        # Post the seam transfer first so it overlaps the local matching.
        reqs, rbuf = self.exchange(feat)
        matches = eng.match_consecutive(feat)
        res = ShardResult(features=feat, matches=matches, first_frame=first_frame)
        for r in reqs:
            r.wait()
        if rbuf is not None:
            b = unpack_boundary(rbuf, feat.cap, self.with_keypoints)
            res.boundary = b
            res.boundary_matches = eng.match_cross(feat, 0, b.desc, b.counts)
        return res
