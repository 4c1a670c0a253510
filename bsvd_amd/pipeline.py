"""Host <-> device pipelining around the hot path (new component; SURVEY.md 8f-4 "frame I/O on device").

The reference moves every frame to the GPU synchronously on the compute stream (``x.cuda()``,
/root/reference/Experimental_root/archs/bsvd_arch.py:520) and brings fp32 results back with ``.cpu()`` after a
``torch.cuda.synchronize()`` (models/validation_seq_infer.py:28, denoising_model.py:187).  Here a sequence of clips
flows through three HIP streams so PCIe never idles the matrix cores:

    upload stream   : pinned uint8 frames -> HBM                      (4x fewer bytes than the reference's fp32 frames)
    compute stream  : bsvd_u8_to_planar -> BSVD forward -> bsvd_planar_to_u8   (all kernels of libbsvd_hip.so)
    download stream : uint8 result -> pinned host buffer

``depth`` slots of pinned staging memory form a ring; slot k is reused once its download has completed.  Results are
yielded in submission order.
"""
import collections

import numpy as np
import torch

from .frame_io import frames_to_input, output_to_frames


class _Slot:
    def __init__(self):
        self.pin_in = self.pin_out = None
        self.dev_in = self.dev_out = None
        self.shape = None
        self.uploaded = torch.cuda.Event()
        self.computed = torch.cuda.Event()
        self.downloaded = torch.cuda.Event()
        self.ticket = None          # the in-flight clip occupying this slot


class _Ticket:
    """One submitted clip: owns its slot until the download has been copied out of the pinned buffer."""

    def __init__(self, slot):
        self.slot, self.result = slot, None

    def finish(self):
        if self.slot is not None:
            self.slot.downloaded.synchronize()
            self.result = self.slot.pin_out.numpy().copy()
            self.slot.ticket = None
            self.slot = None
        return self.result


class ClipPipeline:
    """model: a bsvd_amd.BSVD on a HIP device.  sigma: noise std in [0,1] units for the constant noise map (None for a
    blind model).  depth >= 2 overlaps the transfers of one clip with the forward of another."""

    def __init__(self, model, sigma=None, depth=2):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.model, self.sigma = model, sigma
        self.device = model._device()
        if self.device.type != "cuda":
            raise RuntimeError("ClipPipeline needs the model on a HIP device (model.cuda())")
        with torch.cuda.device(self.device):
            self.up, self.comp, self.down = (torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream())
            self.slots = [_Slot() for _ in range(depth)]
        self.pending = collections.deque()
        self.count = 0

    # ------------------------------------------------------------------------------------------------
    def submit(self, frames_u8):
        """frames_u8: numpy uint8 [T,H,W,3] (RGB).  Enqueues upload + forward + download; returns immediately unless
        the ring is full (then it first drains the oldest clip into its result)."""
        frames_u8 = np.ascontiguousarray(frames_u8)
        if frames_u8.dtype != np.uint8 or frames_u8.ndim != 4 or frames_u8.shape[-1] != 3:
            raise ValueError("expected uint8 frames [T,H,W,3]")
        if frames_u8.shape[1] % 4 or frames_u8.shape[2] % 4:
            raise ValueError("H and W must be multiples of 4 (pad first: denoise.pad_to_multiple_of_4)")
        slot = self.slots[self.count % len(self.slots)]
        self.count += 1
        if slot.ticket is not None:
            slot.ticket.finish()               # ring full: the oldest clip's result leaves the pinned buffer first
        with torch.cuda.device(self.device):
            if slot.shape != frames_u8.shape:
                slot.shape = frames_u8.shape
                slot.pin_in = torch.empty(frames_u8.shape, dtype=torch.uint8).pin_memory()
                slot.pin_out = torch.empty(frames_u8.shape, dtype=torch.uint8).pin_memory()
                slot.dev_in = torch.empty(frames_u8.shape, dtype=torch.uint8, device=self.device)
            slot.pin_in.numpy()[...] = frames_u8
            with torch.cuda.stream(self.up):
                slot.dev_in.copy_(slot.pin_in, non_blocking=True)
                slot.uploaded.record()
            with torch.cuda.stream(self.comp):
                self.comp.wait_event(slot.uploaded)
                x = frames_to_input(slot.dev_in, self.sigma)
                T, _, H, W = x.shape
                if self.model._pick_mode(T, H, W) == "clip":
                    y = self.model.clip_forward(x)
                else:
                    y = self.model.streaming_forward(x)
                slot.dev_out = output_to_frames(y.float())          # held by the slot until its download completed
                slot.computed.record()
            with torch.cuda.stream(self.down):
                self.down.wait_event(slot.computed)
                slot.pin_out.copy_(slot.dev_out, non_blocking=True)
                slot.downloaded.record()
        slot.ticket = _Ticket(slot)
        self.pending.append(slot.ticket)
        return slot.ticket

    def results(self):
        """Drains every clip submitted so far, in submission order."""
        while self.pending:
            yield self.pending.popleft().finish()

    def run(self, clips):
        """clips: iterable of uint8 [T,H,W,3] arrays -> generator of denoised uint8 [T,H,W,3] arrays, in order, with up
        to ``depth`` clips in flight."""
        for clip in clips:
            if len(self.pending) == len(self.slots):
                yield self.pending.popleft().finish()
            self.submit(clip)
        yield from self.results()


class LiveStream:
    """One live feed through the per-frame streaming API (``BSVD.feedin_one_element``, bsvd_arch.py:485-488) with uint8 frames
    on the host side: the streaming counterpart of ``ClipPipeline``.

        feed(frame)  : uint8 [H,W,3] (RGB) -> enqueues upload (uint8 over PCIe), ``bsvd_u8_to_planar``, one pipeline step (a
                       HIP-graph replay), ``bsvd_planar_to_u8`` and the download of whatever that step emitted on three HIP
                       streams, then waits for the step fed ``depth-1`` calls earlier and returns the frame it emitted (None
                       while the 16-step pipeline fills).
        flush()      : feeds the 16 + 1 ``None`` steps of the reference's ``streaming_forward`` tail (:530-544), returns the
                       remaining denoised frames in order and resets the stream.

    Frame k comes back ``model.shift_num`` (16) feeds after it went in -- the network's own latency -- plus ``depth-1``
    feeds of host pipelining (``depth=1``: every feed waits for its own step, lowest latency; ``depth>=2``: transfers of one
    step overlap the compute of the next, highest rate).  ``overlap_blocks`` (default: on for ``depth >= 2``) additionally lets
    DenBlock 2 run one step behind DenBlock 1 as a parallel graph branch (``BSVD.feed_overlapped``): one more feed of latency
    (``shift_num + depth`` in total), the single-frame launches of two independent chains share the chip.  Results keep
    submission order and are byte-identical in every mode.  Not re-entrant (one stream per instance, like the reference's
    module state)."""

    def __init__(self, model, sigma=None, depth=2, overlap_blocks=None, frame_shape=None):
        """frame_shape: optional (H, W) of the frames to come -- the overlap decision (and with it ``latency``) is then final at
        construction instead of at the first feed."""
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.model, self.sigma, self.depth = model, sigma, depth
        self._overlap_wanted = (depth >= 2) if overlap_blocks is None else bool(overlap_blocks)
        self._overlap_explicit = overlap_blocks is not None
        self.overlap = self._overlap_wanted
        self._overlap_decided = not self._overlap_wanted     # nothing to decide without the lagged schedule
        self.device = model._device()
        if self.device.type != "cuda":
            raise RuntimeError("LiveStream needs the model on a HIP device (model.cuda())")
        with torch.cuda.device(self.device):
            self.up, self.comp, self.down = (torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream())
        self.slots, self.shape = None, None
        self.inflight = collections.deque()          # (slot, has_output) in submission order
        self.count = 0
        model.reset()
        if frame_shape is not None:
            self._decide_overlap(int(frame_shape[0]), int(frame_shape[1]))

    @property
    def latency(self):
        """feeds between a frame going in and coming out: ``shift_num + depth - 1``, + 1 with the lagged two-branch schedule.  Final once
        ``latency_final`` is True -- from construction when ``frame_shape`` was given or overlap is off, else from the first feed (a default
        ``overlap_blocks=None`` falls back to the plain per-frame feed when the ring engine is not available for the frame size)."""
        return self.model.shift_num + self.depth - 1 + (1 if self.overlap else 0)

    @property
    def latency_final(self):
        return self._overlap_decided

    def _decide_overlap(self, h, w):
        """The lagged two-branch schedule needs the ring engine (stream_rings, planar edge layers, rings that fit the free HBM).  Decided
        once per stream, BEFORE its first frame touches the pipeline: a default (overlap_blocks=None) falls back to the plain per-frame
        feed -- one feed less latency, same frames --, an explicit overlap_blocks=True raises here with the stream still untouched (and
        the decision still open: a retried feed raises the same error again instead of failing half-way through a step).  flush() and a
        new frame size re-open the decision."""
        if self._overlap_decided:
            return
        if not self.model.overlap_available((self.model.net.net_in_ch, h, w)):
            if self._overlap_explicit:
                raise RuntimeError("LiveStream(overlap_blocks=True) needs the ring engine (stream_rings=True, planar edge layers, "
                                   "enough free HBM for the rings); use overlap_blocks=False")
            self.overlap = False
        self._overlap_decided = True

    def _reopen_overlap(self):
        self.overlap = self._overlap_wanted
        self._overlap_decided = not self._overlap_wanted

    def _alloc(self, shape):
        self.shape = shape
        self.slots = []
        for _ in range(self.depth):                   # step k reuses the slot of step k - depth, which has been handed out
            s = _Slot()
            s.pin_in = torch.empty(shape, dtype=torch.uint8).pin_memory()
            s.pin_out = torch.empty(shape, dtype=torch.uint8).pin_memory()
            s.dev_in = torch.empty((1,) + shape, dtype=torch.uint8, device=self.device)
            self.slots.append(s)

    def _pop(self):
        """oldest in-flight step -> its uint8 frame, or None if that step emitted nothing (pipeline fill)"""
        slot, has_out = self.inflight.popleft()
        slot.downloaded.synchronize()
        return slot.pin_out.numpy().copy() if has_out else None

    def _step(self, frame_u8, last=False):
        if frame_u8 is not None:
            self._decide_overlap(frame_u8.shape[0], frame_u8.shape[1])
        slot = self.slots[self.count % len(self.slots)]
        self.count += 1
        with torch.cuda.device(self.device):
            if frame_u8 is not None:
                slot.pin_in.numpy()[...] = frame_u8
                with torch.cuda.stream(self.up):
                    slot.dev_in[0].copy_(slot.pin_in, non_blocking=True)
                    slot.uploaded.record()
            with torch.cuda.stream(self.comp):
                x = None
                if frame_u8 is not None:
                    self.comp.wait_event(slot.uploaded)
                    x = frames_to_input(slot.dev_in, self.sigma)
                y = self.model.feed_overlapped(x, last=last) if self.overlap else self.model.feedin_one_element(x)
                if y is not None:
                    slot.dev_out = output_to_frames(y.float())      # held by the slot until its download completed
                slot.computed.record()
            with torch.cuda.stream(self.down):
                self.down.wait_event(slot.computed)
                if y is not None:
                    slot.pin_out.copy_(slot.dev_out[0], non_blocking=True)
                slot.downloaded.record()
        self.inflight.append((slot, y is not None))

    def _drain(self, keep, outs):
        while len(self.inflight) > keep:
            r = self._pop()
            if r is not None:
                outs.append(r)

    def feed(self, frame_u8):
        frame_u8 = np.ascontiguousarray(frame_u8)
        if frame_u8.dtype != np.uint8 or frame_u8.ndim != 3 or frame_u8.shape[-1] != 3:
            raise ValueError("expected one uint8 frame [H,W,3]")
        if frame_u8.shape[0] % 4 or frame_u8.shape[1] % 4:
            raise ValueError("H and W must be multiples of 4 (pad first: denoise.pad_to_multiple_of_4)")
        if self.shape != frame_u8.shape:
            if self.inflight:
                raise ValueError("frame size changed mid-stream; flush() first")
            self._reopen_overlap()                    # the rings of another frame size may or may not fit
            self._alloc(frame_u8.shape)
        self._step(frame_u8)                          # step k is in flight ...
        outs = []
        self._drain(self.depth - 1, outs)             # ... while step k - (depth-1) is waited for and handed out
        return outs[0] if outs else None

    def flush(self):
        """end of the feed: the 16 + 1 ``None`` steps of streaming_forward's tail, then everything in flight; returns the
        remaining frames, oldest first, and resets the stream"""
        outs = []
        if self.slots is None:
            return outs
        for _ in range(self.model.shift_num + 1):
            self._step(None)
            self._drain(self.depth - 1, outs)
        if self.overlap:                              # the lagging DenBlock-2 step of the last flush feed
            self._step(None, last=True)
        self._drain(0, outs)
        self.model.reset()
        self._reopen_overlap()                        # the next stream decides again (free HBM may have changed)
        return outs
