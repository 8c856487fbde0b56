"""Frame-range scheduler: contiguous frame ranges per GPU, neighbour exchange of the boundary frame.

The reference processes one video on one thread, matching every frame against the previous one
(reference views.py:75-90).  Extraction is independent per frame and matching couples only frames
f and f-1, so a frame sequence shards into contiguous ranges: rank r owns frames [s_r, e_r), and
the only pair it cannot form by itself is (s_r, s_r - 1), whose train side lives on rank r-1.
Rank r-1 therefore sends the features of its LAST frame to rank r (point-to-point messages to its
direct neighbour; RCCL send/recv when the tensors are on GPUs) and nothing else crosses ranks.

On the wire the boundary frame is a 16-byte header (n, nd, keypoint rows sent, has_keypoints) followed
by exactly the live rows: nd descriptors of 32 bytes and, optionally, min(n, cap) keypoint rows of 28
bytes, sent straight out of the feature arrays (no packing copy).  The sender extracts its last frame
FIRST, learns the two counts from a 8-byte copy to pinned host memory, and posts the sends on a side
stream while the rest of its range is still being extracted; the receiver posts the header receive
before it starts its own work and the row receives once the header has told it their sizes.  Both
host waits happen after the rank's own GPU work has been enqueued, so the GPU never idles for them.

One process per GPU; torch.distributed provides the transport: "nccl" (= RCCL) moves device tensors
directly; with "gloo" device tensors are staged through host memory (used to rehearse several ranks
on one GPU, where RCCL refuses to run), and CPU tensors (tests of the host logic) go as they are.
"""
from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist

HEADER_BYTES = 16  # int32 n, nd, keypoint rows, has_keypoints


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
    """Size of the capacity-padded single-buffer form of a boundary frame (pack_boundary)."""
    return HEADER_BYTES + cap * 32 + (cap * 28 if with_keypoints else 0)


def boundary_wire_bytes(n, nd, cap, with_keypoints=True):
    """Bytes that actually travel for a frame with counts (n, nd): header + live rows."""
    return HEADER_BYTES + min(nd, cap) * 32 + (min(n, cap) * 28 if with_keypoints else 0)


def pack_boundary(feat, frame, with_keypoints=True):
    """Features of one frame -> one contiguous capacity-padded uint8 buffer on the same device (the layout
    the receiver keeps a boundary frame in; see unpack_boundary)."""
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
    """A received boundary frame, as views into one capacity-sized buffer (no copies)."""
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
    boundary: Optional[Boundary] = None      # views into the scheduler's receive buffer: valid until its next process()
    first_frame: int = 0
    wire_bytes_sent: int = 0         # header + live rows sent to the next rank
    wire_bytes_received: int = 0


class _Wire:
    """Point-to-point transport of tensors between neighbouring ranks.  Device tensors travel as they are
    over a backend that moves device memory (nccl = RCCL), through host copies otherwise."""

    def __init__(self, group, device):
        self.group = group
        self.on_gpu = device.type == "cuda"
        backend = dist.get_backend(group) if dist.is_initialized() else "gloo"
        self.staged = self.on_gpu and "nccl" not in str(backend)
        self.device = device

    def recv(self, tensors, src):
        """Receive into contiguous tensors (views of the destination buffer); blocks until they have arrived
        and, for staged transfers, copies them to the device."""
        if not tensors:
            return
        bufs = [torch.empty(t.shape, dtype=t.dtype) if self.staged else t for t in tensors]
        for r in dist.batch_isend_irecv([dist.P2POp(dist.irecv, b, src, self.group) for b in bufs]):
            r.wait()
        if self.staged:
            for t, b in zip(tensors, bufs):
                t.copy_(b)


class FrameRangeScheduler:
    """Runs one rank's share of a frame sequence and stitches the seam to the previous rank.

    `engine` provides alloc_features(frames), extract_batch(gray, out=), match_consecutive(features, out=)
    and match_cross(features, qframe, t_desc, t_counts, out=) (visual_slam_amd.engine.OrbEngine).
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
        self._side = None           # side stream for the seam transfers (device tensors only)
        self._host_counts = None    # pinned [2] int32
        self._rbuf = None           # capacity-sized receive buffer, reused from step to step

    # -- helpers ------------------------------------------------------------------------------
    def _streams(self, device):
        """(main stream, side stream) for device tensors, (None, None) on the CPU."""
        if device.type != "cuda":
            return None, None
        if self._side is None:
            self._side = torch.cuda.Stream(device)
        return torch.cuda.current_stream(device), self._side

    def _receive_boundary(self, wire, device, cap, hdr_req, hdr_buf, side):
        """Header (already posted) -> sizes -> live rows into a capacity-sized buffer.  Returns
        (Boundary, wire bytes)."""
        ctx = torch.cuda.stream(side) if side is not None else _null_ctx()
        with ctx:
            for r in hdr_req:
                r.wait()
            hdr = [int(v) for v in hdr_buf.cpu().tolist()]
            n, nd, nk, haskp = hdr
            if nd < 0 or nk < 0 or nd > cap or nk > cap or bool(haskp) != self.with_keypoints:
                raise RuntimeError("boundary header does not fit this rank's buffers: n=%d nd=%d rows=%d "
                                   "keypoints=%d (capacity %d)" % (n, nd, nk, haskp, cap))
            nbytes = boundary_message_bytes(cap, self.with_keypoints)
            if self._rbuf is None or self._rbuf.numel() != nbytes or self._rbuf.device != device:
                self._rbuf = torch.empty(nbytes, dtype=torch.uint8, device=device)
            b = unpack_boundary(self._rbuf, cap, self.with_keypoints)
            b.counts.copy_(torch.tensor([[n, nd]], dtype=torch.int32), non_blocking=False)
            parts = []
            if nd > 0:
                parts.append(b.desc[0, :nd])
            if self.with_keypoints and nk > 0:
                parts.append(b.kp7[0, :nk])
            wire.recv(parts, self.rank - 1)
        return b, (n, nd, nk), HEADER_BYTES + nd * 32 + (nk * 28 if self.with_keypoints else 0)

    def _post_header_recv(self, wire, device, side):
        hdr_buf = torch.empty(4, dtype=torch.int32, device="cpu" if wire.staged else device)
        ctx = torch.cuda.stream(side) if (side is not None and not wire.staged) else _null_ctx()
        with ctx:
            reqs = dist.batch_isend_irecv([dist.P2POp(dist.irecv, hdr_buf, self.rank - 1, self.group)])
        return reqs, hdr_buf

    def _send_boundary(self, wire, n, nd, nk, desc_rows, kp_rows, device):
        """Header + live rows to rank + 1.  Returns (requests, the tensors in flight, wire bytes)."""
        hdr = torch.tensor([n, nd, nk, 1 if self.with_keypoints else 0], dtype=torch.int32,
                           device="cpu" if wire.staged else device)
        parts = [hdr]
        if nd > 0:
            parts.append(desc_rows)
        if self.with_keypoints and nk > 0:
            parts.append(kp_rows)
        if wire.staged:
            parts = [hdr] + [p.cpu() for p in parts[1:]]
        # two batches, as the receiver posts them: the header alone, then the rows
        reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, hdr, self.rank + 1, self.group)])
        if len(parts) > 1:
            reqs += dist.batch_isend_irecv([dist.P2POp(dist.isend, p, self.rank + 1, self.group) for p in parts[1:]])
        return reqs, parts, HEADER_BYTES + nd * 32 + (nk * 28 if self.with_keypoints else 0)

    def _relay(self, device):
        """A rank without frames keeps the seam chain intact: it passes the previous rank's boundary
        frame on to the next rank unchanged (rank 0 without frames sends an empty frame), so its
        neighbours' transfers complete and the next rank that has frames still matches against the
        last frame that exists."""
        wire = _Wire(self.group, device)
        cap = self.engine.cap
        n = nd = nk = 0
        b = None
        if self.rank > 0:
            reqs, hdr_buf = self._post_header_recv(wire, device, None)
            b, (n, nd, nk), _ = self._receive_boundary(wire, device, cap, reqs, hdr_buf, None)
        if self.rank + 1 < self.world:
            desc_rows = b.desc[0, :nd] if b is not None else None
            kp_rows = b.kp7[0, :nk] if (b is not None and self.with_keypoints) else None
            reqs, keep, _ = self._send_boundary(wire, n, nd, nk, desc_rows, kp_rows, device)
            for r in reqs:
                r.wait()
            del keep
        return b

    # -- one step ---------------------------------------------------------------------------------
    def process(self, gray, first_frame=0, out=None, matches=None, seam=None):
        """gray: this rank's frames [frames, H, W].  Returns a ShardResult; all GPU work is enqueued
        asynchronously where the engine is asynchronous.  `out`, `matches`, `seam`: preallocated
        BatchFeatures / BatchMatches to write into (allocated here when None).  A rank whose range is
        empty (more ranks than frames) returns a ShardResult without features."""
        eng = self.engine
        F = gray.shape[0]
        device = gray.device
        if F == 0:
            if self.world > 1:
                self._relay(device)
            return ShardResult(features=None, matches=None, first_frame=first_frame)
        feat = out if out is not None else eng.alloc_features(F)
        cap = feat.cap
        sending = self.world > 1 and self.rank + 1 < self.world
        receiving = self.world > 1 and self.rank > 0
        res = ShardResult(features=feat, matches=None, first_frame=first_frame)
        if not (sending or receiving):
            eng.extract_batch(gray, out=feat)
            res.matches = eng.match_consecutive(feat, out=matches)
            return res

        wire = _Wire(self.group, device)
        main, side = self._streams(device)
        if side is not None:
            side.wait_stream(main)       # the previous step's seam match may still be reading the receive buffer
        hdr_reqs = hdr_buf = None
        if receiving:
            hdr_reqs, hdr_buf = self._post_header_recv(wire, device, side)

        send_reqs, keep, ev = [], None, None
        if sending:
            # the boundary frame first: its transfer overlaps everything else this rank does
            eng.extract_batch(gray[F - 1:F], out=feat.view(F - 1, F))
            if main is not None:
                if self._host_counts is None:
                    self._host_counts = torch.empty(2, dtype=torch.int32).pin_memory()
                self._host_counts.copy_(feat.counts[F - 1], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(main)
            if F > 1:
                eng.extract_batch(gray[:F - 1], out=feat.view(0, F - 1))
        else:
            eng.extract_batch(gray, out=feat)
        res.matches = eng.match_consecutive(feat, out=matches)

        if sending:
            if ev is not None:
                ev.synchronize()                      # the boundary frame's counts are on the host
                n, nd = (int(v) for v in self._host_counts.tolist())
            else:
                n, nd = (int(v) for v in feat.counts[F - 1].tolist())
            nk, nd = min(n, cap), min(nd, cap)
            ctx = torch.cuda.stream(side) if side is not None else _null_ctx()
            with ctx:
                if side is not None:
                    side.wait_event(ev)
                send_reqs, keep, res.wire_bytes_sent = self._send_boundary(
                    wire, n, nd, nk, feat.desc[F - 1, :nd], feat.kp7[F - 1, :nk] if self.with_keypoints else None,
                    device)

        if receiving:
            b, _, res.wire_bytes_received = self._receive_boundary(wire, device, cap, hdr_reqs, hdr_buf, side)
            if main is not None:
                main.wait_stream(side)
            res.boundary = b
            res.boundary_matches = eng.match_cross(feat, 0, b.desc, b.counts, out=seam)
        for r in send_reqs:
            r.wait()
        if sending and main is not None and not wire.staged:
            main.wait_stream(side)                    # the sent rows must not be overwritten by the next step
        del keep
        return res


class _null_ctx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
