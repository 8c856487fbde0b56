"""Frame-batch engine: device-resident extraction and matching on one GPU.

PyTorch is used only as plumbing here (device memory, the stream, and later torch.distributed);
every computation goes through the C ABI of liborbamd.so into the HIP kernels.
"""
from dataclasses import dataclass

import torch

from . import _lib, synth


@dataclass
class BatchFeatures:
    """Frame-strided outputs of one extract_batch call (all on the GPU)."""
    kp7: torch.Tensor        # float32 [frames, cap, 7]
    desc: torch.Tensor       # uint8   [frames, cap, 32]
    valid_idx: torch.Tensor  # int32   [frames, cap]
    counts: torch.Tensor     # int32   [frames, 2] = (n, nd): keypoints found (may exceed cap), descriptor rows stored
    cap: int

    @property
    def frames(self):
        return self.counts.shape[0]

    def frame(self, f):
        """Host copies of one frame, trimmed to its counts: (kp [n,7], desc [nv,32], valid_idx [nv])."""
        n, nv = (int(v) for v in self.counts[f].tolist())
        n, nv = min(n, self.cap), min(nv, self.cap)
        return (self.kp7[f, :n].cpu().numpy(), self.desc[f, :nv].cpu().numpy(),
                self.valid_idx[f, :nv].cpu().numpy())

    def overflowed(self):
        """Indices of the frames that hold more keypoints than `cap` (synchronises).  Such a frame
        keeps its first cap keypoints and their descriptors; matching uses exactly those rows."""
        return torch.nonzero(self.counts[:, 0] > self.cap).flatten().tolist()

    def view(self, a, b):
        """Frames [a, b) as views (no copies)."""
        return BatchFeatures(kp7=self.kp7[a:b], desc=self.desc[a:b], valid_idx=self.valid_idx[a:b],
                             counts=self.counts[a:b], cap=self.cap)


@dataclass
class BatchMatches:
    """Top-2 neighbours for a list of frame pairs (all on the GPU)."""
    idx: torch.Tensor    # int32 [npairs, cap, 2]
    dist: torch.Tensor   # int32 [npairs, cap, 2]
    good: torch.Tensor   # uint8 [npairs, cap]   Lowe ratio 0.75 passed
    cap: int

    def pair(self, p, nq):
        return (self.idx[p, :nq].cpu().numpy(), self.dist[p, :nq].cpu().numpy(),
                self.good[p, :nq].cpu().numpy())

    def view(self, a, b):
        """Pairs [a, b) as views (no copies)."""
        return BatchMatches(idx=self.idx[a:b], dist=self.dist[a:b], good=self.good[a:b], cap=self.cap)


class OrbEngine:
    """One GPU's worth of the ORB front-end for frame batches resident in device memory."""

    def __init__(self, device=0, cap=8192):
        if not torch.cuda.is_available():
            raise RuntimeError("OrbEngine needs a GPU: torch.cuda.is_available() is False and this "
                               "package has no CPU path")
        self.device = torch.device("cuda", device)
        self.cap = int(cap)
        torch.cuda.set_device(self.device)
        self.ctx = _lib.Context(device)
        self.bind_stream()
        self._match_ctx = None        # second context + stream, created by extract_and_match()
        self._match_stream = None
        self._matcher = _lib.MATCHER_MATRIX

    def bind_stream(self, stream=None):
        """Run the library's kernels on a torch stream (default: the current one), so that tensor
        allocation, torch.distributed and the HIP kernels are ordered on one queue."""
        stream = stream or torch.cuda.current_stream(self.device)
        self.stream = stream
        self.ctx.set_stream(stream.cuda_stream)

    def sync(self):
        self.ctx.sync()
        if self._match_ctx is not None:
            self._match_ctx.sync()

    def set_matcher(self, matcher):
        """_lib.MATCHER_MATRIX (default) or _lib.MATCHER_VALU, for every matching call of this engine."""
        self._matcher = matcher
        self.ctx.set_matcher(matcher)
        if self._match_ctx is not None:
            self._match_ctx.set_matcher(matcher)

    def set_audit(self, on=True):
        """Parity audit of the following extract_batch calls (see orbamd_ctx_set_audit in include/orbamd.h)."""
        self.ctx.set_audit(on)

    def audit_read(self, frames):
        """int32 [frames, 4] numpy array for the last extract_batch call (synchronises)."""
        return self.ctx.audit_read(frames)

    # -- inputs ---------------------------------------------------------------------------------
    def synth_frames(self, frames, H, W, first_frame=0, seed=synth.DEFAULT_SEED, cell=20, density=0.386):
        out = torch.empty((frames, H, W), dtype=torch.uint8, device=self.device)
        self.ctx.synth_frames_dev(out.data_ptr(), frames, H, W, first_frame, seed, cell,
                                  synth.density_threshold(density))
        return out

    def synth_textured_frames(self, frames, H, W, first_frame=0, seed=synth.DEFAULT_SEED):
        out = torch.empty((frames, H, W), dtype=torch.uint8, device=self.device)
        self.ctx.synth_textured_dev(out.data_ptr(), frames, H, W, first_frame, seed)
        return out

    def bgr2gray(self, bgr):
        assert bgr.dtype == torch.uint8 and bgr.dim() == 4 and bgr.shape[3] == 3 and bgr.is_contiguous()
        frames, H, W = bgr.shape[:3]
        out = torch.empty((frames, H, W), dtype=torch.uint8, device=self.device)
        self.ctx.bgr2gray_dev(bgr.data_ptr(), frames, H, W, out.data_ptr())
        return out

    # -- the hot path ---------------------------------------------------------------------------
    def alloc_features(self, frames, cap=None):
        cap = cap or self.cap
        dev = self.device
        return BatchFeatures(
            kp7=torch.empty((frames, cap, 7), dtype=torch.float32, device=dev),
            desc=torch.empty((frames, cap, 32), dtype=torch.uint8, device=dev),
            valid_idx=torch.empty((frames, cap), dtype=torch.int32, device=dev),
            counts=torch.empty((frames, 2), dtype=torch.int32, device=dev), cap=cap)

    def alloc_matches(self, npairs, cap=None):
        cap = cap or self.cap
        dev = self.device
        return BatchMatches(
            idx=torch.empty((max(npairs, 1), cap, 2), dtype=torch.int32, device=dev),
            dist=torch.empty((max(npairs, 1), cap, 2), dtype=torch.int32, device=dev),
            good=torch.empty((max(npairs, 1), cap), dtype=torch.uint8, device=dev), cap=cap)

    def extract_batch(self, gray, out=None):
        """detect + orient + describe for gray uint8 [frames, H, W] on this GPU (asynchronous)."""
        assert gray.dtype == torch.uint8 and gray.dim() == 3 and gray.is_contiguous() and gray.is_cuda
        frames, H, W = gray.shape
        out = out or self.alloc_features(frames)
        assert out.frames == frames
        self.ctx.extract_batch_dev(gray.data_ptr(), frames, H, W, out.cap, out.kp7.data_ptr(),
                                   out.desc.data_ptr(), out.valid_idx.data_ptr(), out.counts.data_ptr())
        return out

    def match_consecutive(self, feat, out=None):
        """Pairs (f, f-1) for f = 1 .. frames-1: current frame queries, previous frame trains."""
        npairs = feat.frames - 1
        out = out or self.alloc_matches(npairs, feat.cap)
        if npairs > 0:
            self.ctx.knn2_batch_dev(feat.desc.data_ptr(), feat.counts.data_ptr(), feat.frames, feat.cap, 0,
                                    npairs, out.idx.data_ptr(), out.dist.data_ptr(), out.good.data_ptr())
        return out

    def extract_and_match(self, gray, feat, matches, chunks=4):
        """extract_batch(gray) + match_consecutive(feat), pipelined: the frames are cut into `chunks`
        ranges; while range k+1 is extracted on this engine's stream, range k (plus the pair that
        joins it to range k-1) is matched on a second stream by a second library context.  The
        extraction kernels are bound by vector-ALU, LDS and load-address work, the matcher by the
        matrix cores, so the two overlap on the same compute units.  Results are those of the two
        separate calls.  On return the engine's stream waits for the matching stream."""
        frames = gray.shape[0]
        assert feat.frames == frames and matches.idx.shape[0] >= max(frames - 1, 1)
        chunks = max(1, min(int(chunks), frames))
        if chunks == 1 or frames < 2:
            self.extract_batch(gray, out=feat)
            return self.match_consecutive(feat, out=matches)
        if self._match_ctx is None:
            self._match_stream = torch.cuda.Stream(self.device)
            self._match_ctx = _lib.Context(self.device.index)
            self._match_ctx.set_stream(self._match_stream.cuda_stream)
            self._match_ctx.set_matcher(self._matcher)
            self._chunk_events = []
        while len(self._chunk_events) < chunks:
            self._chunk_events.append(torch.cuda.Event())
        main, side = self.stream, self._match_stream
        side.wait_stream(main)                      # earlier work on the outputs is ordered first
        for k in range(chunks):
            lo, hi = frames * k // chunks, frames * (k + 1) // chunks
            self.extract_batch(gray[lo:hi], out=feat.view(lo, hi))
            self._chunk_events[k].record(main)
            side.wait_event(self._chunk_events[k])
            q0 = max(lo - 1, 0)                     # the pair (lo, lo - 1) belongs to this range
            if hi - q0 >= 2:
                f = feat.view(q0, hi)
                o = matches.view(q0, hi - 1)
                self._match_ctx.knn2_batch_dev(f.desc.data_ptr(), f.counts.data_ptr(), f.frames, f.cap, 0,
                                               f.frames - 1, o.idx.data_ptr(), o.dist.data_ptr(),
                                               o.good.data_ptr())
        main.wait_stream(side)
        return matches

    def good_matches(self, feat, matches, pairs=None):
        """The reference's ratio-filter lists (views.py:163-171) for every pair of `matches`, built on
        the GPU: (good_q, good_t2, good_t1) int32 [npairs, cap] and kept int32 [npairs]; row p holds
        kept[p] entries (m1.queryIdx, m2.trainIdx, m1.trainIdx), kept[p] = -1 if a query of the pair
        has fewer than two neighbours.  `pairs` as given to match_pairs, None for match_consecutive."""
        npairs = pairs.shape[0] if pairs is not None else feat.frames - 1
        cap, dev = matches.cap, self.device
        gq = torch.empty((max(npairs, 1), cap), dtype=torch.int32, device=dev)
        gt2 = torch.empty_like(gq)
        gt1 = torch.empty_like(gq)
        kept = torch.zeros((max(npairs, 1),), dtype=torch.int32, device=dev)
        if npairs > 0:
            self.ctx.ratio_filter_batch_dev(matches.idx.data_ptr(), matches.good.data_ptr(), feat.counts.data_ptr(),
                                            feat.frames, cap, pairs.data_ptr() if pairs is not None else 0, npairs,
                                            gq.data_ptr(), gt2.data_ptr(), gt1.data_ptr(), kept.data_ptr())
        return gq, gt2, gt1, kept

    def match_pairs(self, feat, pairs, out=None):
        """Arbitrary (query frame, train frame) pairs inside one batch; pairs: int32 [npairs, 2] on GPU."""
        npairs = pairs.shape[0]
        out = out or self.alloc_matches(npairs, feat.cap)
        self.ctx.knn2_batch_dev(feat.desc.data_ptr(), feat.counts.data_ptr(), feat.frames, feat.cap,
                                pairs.data_ptr(), npairs, out.idx.data_ptr(), out.dist.data_ptr(),
                                out.good.data_ptr())
        return out

    def match_cross(self, feat, qframe, t_desc, t_counts, tframe=0, out=None):
        """Query frame `qframe` of `feat` against frame `tframe` of another frame-strided buffer
        (t_desc uint8 [*, cap, 32], t_counts int32 [*, 2]) -- the pair that spans two GPUs."""
        out = out or self.alloc_matches(1, feat.cap)
        assert t_desc.shape[1] == feat.cap
        self.ctx.knn2_cross_dev(feat.desc.data_ptr(), feat.counts.data_ptr(), qframe, t_desc.data_ptr(),
                                t_counts.data_ptr(), tframe, feat.cap, out.idx.data_ptr(), out.dist.data_ptr(),
                                out.good.data_ptr())
        return out

    def match_sets(self, q, t):
        """Plain descriptor sets uint8 [nq, 32] / [nt, 32] on this GPU -> (idx, dist, good) tensors."""
        nq, nt = q.shape[0], t.shape[0]
        idx = torch.empty((nq, 2), dtype=torch.int32, device=self.device)
        dist = torch.empty((nq, 2), dtype=torch.int32, device=self.device)
        good = torch.empty((nq,), dtype=torch.uint8, device=self.device)
        self.ctx.knn2_dev(q.data_ptr(), nq, t.data_ptr() if nt else 0, nt, idx.data_ptr(), dist.data_ptr(),
                          good.data_ptr())
        return idx, dist, good
