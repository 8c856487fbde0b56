"""Frame staging for video that starts in host memory: pinned ring buffer -> device on a copy stream.

The reference reads one frame at a time from cv2.VideoCapture on its worker thread and hands it to
orb.extract (reference views.py:75-90, :273).  Here a decoder (or any producer of frames) writes into
pinned host chunks; the copy of chunk k+1 runs on its own stream while chunk k goes through
extraction and matching, so the PCIe link and the kernels overlap.  Frames may be gray [H, W] or
BGR [H, W, 3] uint8 (converted on the GPU with the formula of reference src/main_interface.cpp:83).

Matching follows the reference: every frame is matched against the previous one, across chunk
boundaries too -- each chunk's feature buffer keeps the previous chunk's last frame in slot 0.
"""
from dataclasses import dataclass

import numpy as np
import torch

from .engine import BatchFeatures, BatchMatches


@dataclass
class StagedChunk:
    """Results for frames [first, first + frames) of the stream (device tensors owned by the stager and
    overwritten two chunks later: consume or copy them before asking for the chunk after the next)."""
    first: int
    frames: int
    features: BatchFeatures     # this chunk's frames only
    matches: BatchMatches       # pair p = (frame first + p) against (frame first + p - 1); for the very
                                # first frame of the stream pair 0 has no train side (all -1)


class HostFrameStager:
    def __init__(self, engine, H, W, chunk=64, bgr=False):
        self.eng, self.H, self.W, self.chunk, self.bgr = engine, int(H), int(W), int(chunk), bool(bgr)
        shape = (self.chunk, self.H, self.W, 3) if bgr else (self.chunk, self.H, self.W)
        dev = engine.device
        self.host = [torch.empty(shape, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.dev_in = [torch.empty(shape, dtype=torch.uint8, device=dev) for _ in range(2)]
        self.copy_stream = torch.cuda.Stream(dev)
        self.copied = [torch.cuda.Event() for _ in range(2)]
        self.consumed = [torch.cuda.Event() for _ in range(2)]
        # slot 0 = last frame of the previous chunk
        self.feat = [engine.alloc_features(self.chunk + 1) for _ in range(2)]
        self.match = [engine.alloc_matches(self.chunk) for _ in range(2)]
        for f in self.feat:
            f.counts.zero_()
        for e in self.consumed:
            e.record(torch.cuda.current_stream(dev))

    def _upload(self, k, frames):
        """Host chunk k (already written into self.host[k & 1][:frames]) -> device, on the copy stream."""
        b = k & 1
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(self.consumed[b])        # the kernels that read this buffer are done
            self.dev_in[b][:frames].copy_(self.host[b][:frames], non_blocking=True)
            self.copied[b].record(self.copy_stream)

    def _process(self, k, first, frames):
        eng, b = self.eng, k & 1
        main = torch.cuda.current_stream(eng.device)
        main.wait_event(self.copied[b])
        src = self.dev_in[b][:frames]
        gray = eng.bgr2gray(src) if self.bgr else src
        feat, prev = self.feat[b], self.feat[b ^ 1]
        if k > 0:                                                 # previous chunk's last frame -> slot 0
            pl = self._last_frames
            feat.desc[0].copy_(prev.desc[pl])
            feat.kp7[0].copy_(prev.kp7[pl])
            feat.valid_idx[0].copy_(prev.valid_idx[pl])
            feat.counts[0].copy_(prev.counts[pl])
        view = feat.view(0, frames + 1)
        new = feat.view(1, frames + 1)
        eng.extract_batch(gray, out=new)
        eng.match_consecutive(view, out=self.match[b].view(0, frames))
        self.consumed[b].record(main)
        self._last_frames = frames
        return StagedChunk(first=first, frames=frames, features=new, matches=self.match[b].view(0, frames))

    def run(self, frames_iter):
        """frames_iter yields uint8 arrays [H, W] (or [H, W, 3] when bgr) -- or is a frame source with
        read_into(array) -> bool (visual_slam_amd.frames.FrameFile), which then reads every frame straight
        into the pinned host chunk, without an intermediate copy.  Yields a StagedChunk per `chunk` frames
        (the last one may be shorter), one chunk behind the uploads."""
        k, fill, first = 0, 0, 0
        pending = None          # (k, first, frames) uploaded but not processed
        for f in self.feat:
            f.counts[0].zero_()  # a new stream has no previous frame
        read_into = getattr(frames_iter, "read_into", None)
        it = None if read_into else iter(frames_iter)
        while True:
            b = k & 1
            if fill == 0:
                # the pinned chunk is free once its previous upload has finished
                self.copied[b].synchronize() if k >= 2 else None
            if read_into:
                if not read_into(self.host[b][fill].numpy()):
                    break
            else:
                try:
                    fr = next(it)
                except StopIteration:
                    break
                self.host[b][fill].copy_(torch.from_numpy(np.ascontiguousarray(fr)))
            fill += 1
            if fill == self.chunk:
                self._upload(k, fill)
                if pending is not None:
                    yield self._process(*pending)
                pending = (k, first, fill)
                first += fill
                k, fill = k + 1, 0
        if fill:
            self._upload(k, fill)
            if pending is not None:
                yield self._process(*pending)
            pending = (k, first, fill)
        if pending is not None:
            yield self._process(*pending)
