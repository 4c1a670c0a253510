"""The stream schedule as fixed rings + per-step launch plans + HIP-graph replay.

``schedule.StreamPipeline`` is the reference's frame-major pipeline (BiBufferConv frame buffers, MemSkip FIFOs, the
None-in/None-out protocol; /root/reference/Experimental_root/archs/bsvd_arch.py:53-114, 308-322, 374-396, 485-488).
Driving it with ``HipExecutor`` costs one ``torch.empty`` and one ctypes launch per layer, 32 per frame.  This module
drives the SAME state machine with a recording executor instead:

* every layer output lives in a preallocated **ring** of device buffers whose depth covers the tensor's lifetime in the
  pipeline (3 productions for the input of a temporal-fusion conv: next / pending / past frame buffer; 5 resp. 9 for the
  skip FIFOs; +1 on the hand-over between the two DenBlocks so that they may run one step apart).  Nothing is allocated
  per step except the tensor handed to the caller;
* a step of one DenBlock becomes a **plan**: the ordered ``BsvdConvArgs`` of the layers that are active in that step,
  identified by a signature (layer, frames, buffer addresses).  In the steady state the signature repeats with the ring
  period (10 steps); pipeline fill / flush steps repeat from clip to clip;
* a plan seen for the second time is captured into a **HIP graph** (``bsvd_graph_*``, include/bsvd_hip.h) and replayed
  with one call per step afterwards; until then it is issued with one ``bsvd_conv3x3_batch`` call.  For
  ``streaming_forward`` (the whole list in hand) DenBlock 1 of step k and DenBlock 2 of step k-1 are two parallel
  branches of one graph -- the single-frame launches of the quarter-resolution layers do not fill 256 CUs on their own;
* with the whole list in hand a step may also carry a **chunk** of n consecutive frames instead of one (ring slots are
  [n,H,W,C]; a temporal-fusion conv reads t-1 / t+1 inside the chunk and takes the previous chunk's last / the next
  chunk's first frame as halos): the same 16-step pipeline at n frames per step -- the clip schedule's launch
  efficiency at O(n) memory instead of O(clip length).

Same kernels, same arguments, same order per buffer as the allocating path: results are bit-identical (tested).
The host logic is executor-agnostic: with a plain executor (tests/oracle_exec.py on CPU) the plans are run layer by
layer, which is how the ring lifetimes are tested without a GPU.
"""
import ctypes
from collections import OrderedDict

import torch

from .schedule import _DenBlockStream

RING_PERIOD = 10            # every ring depth divides it -> the steady-state signature has period 10
_DEPTHS = (1, 2, 5, 10)

# lifetime (in productions, the producing step included) of each layer's output inside one DenBlock
_TSM_FEEDERS = ("down0", "d0c1", "down1", "d1c1", "d1c2", "u2c1", "up2", "u1c1")    # -> BiBufferConv next/pending/past
_LIFETIME = {"inc3": 9, "d0c2": 5}                                                   # skip FIFOs x0 (8 steps), x1 (4 steps)


def ring_depth(name, is_handover=False, is_exit=False):
    need = 3 if name in _TSM_FEEDERS else _LIFETIME.get(name, 1)
    if is_handover:          # temp1.out3 = temp2's input and residual base (8 steps) + 1: the blocks may run one step apart
        need = 10
    if is_exit:              # the caller copies the result out right after the step; 2: the copy may trail the next step
        need = 2
    return min(d for d in _DEPTHS if d >= need)


def ring_bytes_estimate(net, H, W, chunk=1, v_keys=()):
    """Device bytes StreamEngine(net, ..., H, W, chunk) allocates for its rings (fp32 words; split16 is the same size; the layers in
    ``v_keys`` write the transformed domain: (m + 2) / m = 4 / 3 of the pixels' bytes for F(6,3), group padding and the edge record on top)."""
    total = 10 * chunk * net.net_in_ch * H * W
    for blk in (net.temp1, net.temp2):
        h, w = H, W
        for name, sp in blk.items():
            ho, wo = (h - 1) // sp.stride + 1, (w - 1) // sp.stride + 1
            if sp.epilogue == 1:
                n, h, w = 4 * ho * wo * (sp.cout_pad // 4), 2 * ho, 2 * wo
            else:
                n, h, w = ho * wo * sp.cout_pad, ho, wo
            exit_ = blk is net.temp2 and name == "out3"
            if exit_:
                n = sp.cout * ho * wo
            if sp.key in v_keys:
                n = int(n * 1.45)
            total += n * chunk * ring_depth(name, blk is net.temp1 and name == "out3", exit_)
    return 4 * total


class _Recorder:
    """Executor facade for ``_DenBlockStream.feed``: assigns ring slots and records the launches of one step."""
    planar_io = True

    def __init__(self, eng):
        self.eng = eng
        self.count = {}
        self.rec = []

    def reset(self):
        self.count.clear()
        self.rec = []

    def conv(self, sp, x, halo_prev=None, halo_next=None, extra=None, extra_pstride=0, extra_cstride=1,
             x_planar=False, y_planar=None, out=None):
        ring = self.eng.rings[sp.key]
        n = self.count.get(sp.key, 0)
        self.count[sp.key] = n + 1
        o = ring[n % len(ring)]
        T = x.shape[0]
        if T != o.shape[0]:
            o = o[:T]                 # the last chunk of a clip may be shorter
        self.rec.append((sp, x, halo_prev, halo_next, extra, extra_pstride, extra_cstride, x_planar, y_planar, o, None, None))
        return o

    def fuse_head(self, S):
        fn = getattr(self.eng.ex, "fuse_head", None)
        return bool(fn and fn(S))

    def fuse_pair(self, S, na, nb):
        fn = getattr(self.eng.ex, "fuse_pair", None)
        return bool(fn and fn(S, na, nb))

    def conv_pair_fused(self, spa, spb, x, extra=None, extra_pstride=0, extra_cstride=1, y_planar=None, out=None):
        """a fused pair of plain convs as ONE recorded launch writing spb's ring (spa's ring does not exist)"""
        ring = self.eng.rings[spb.key]
        n = self.count.get(spb.key, 0)
        self.count[spb.key] = n + 1
        o = ring[n % len(ring)]
        if x.shape[0] != o.shape[0]:
            o = o[:x.shape[0]]
        self.rec.append((spb, x, None, None, extra, extra_pstride, extra_cstride, False, y_planar, o, None, spa))
        return o

    def conv_head_fused(self, sp0, sp3, x, out=None):
        """the fused entry pair as ONE recorded launch writing inc3's ring (inc0's ring stays unused)"""
        ring = self.eng.rings[sp3.key]
        n = self.count.get(sp3.key, 0)
        self.count[sp3.key] = n + 1
        o = ring[n % len(ring)]
        if x.shape[0] != o.shape[0]:
            o = o[:x.shape[0]]
        self.rec.append((sp3, x, None, None, None, 0, 1, True, None, o, sp0, None))
        return o

    def take(self):
        rec, self.rec = self.rec, []
        return rec


def _signature(rec):
    return tuple((r[0].key, r[1].shape[0], r[1].data_ptr(), r[2].t.data_ptr() if r[2] is not None else 0,
                  r[3].t.data_ptr() if r[3] is not None else 0, r[4].data_ptr() if r[4] is not None else 0,
                  r[9].data_ptr()) for r in rec)


class _Plan:
    __slots__ = ("rec", "args", "n")

    def __init__(self, rec):
        self.rec, self.args, self.n = rec, None, len(rec)


class StreamEngine:
    """feedin_one_element / streaming_forward on fixed rings for one (network, frame size, chunk, device, arithmetic mode)."""

    def __init__(self, net, ex, H, W, in_ch, chunk=1, alloc=None, use_graphs=None, max_graphs=1024, poison=False):
        self.net, self.ex, self.H, self.W, self.chunk = net, ex, H, W, int(chunk)
        self.hip = hasattr(ex, "lib") and hasattr(ex, "build_args")
        self.use_graphs = self.hip if use_graphs is None else (bool(use_graphs) and self.hip)
        if alloc is None:
            dev = ex.device
            alloc = lambda shape: torch.empty(shape, dtype=torch.float32, device=dev)      # noqa: E731
        self.rings = {}
        self.ring_bytes = 0
        n = self.chunk

        out_v = getattr(ex, "out_v", None)

        def ring(key, shape, depth, vm=0):
            if vm:                    # a layer that writes the transformed domain (engine.VT): [n, H, W, C] logical, its own frame size
                from .engine import VT
                buf = alloc((depth, shape[0], VT.frame_elems(shape[1], shape[2], shape[3], vm)))
            else:
                buf = alloc((depth,) + tuple(shape))
            if poison:                # tests: a slot read before it was written shows up as NaN
                buf.fill_(float("nan"))
            self.ring_bytes += buf.numel() * 4
            self.rings[key] = [VT(buf[i], shape[1], shape[2], shape[3], vm) if vm else buf[i] for i in range(depth)]

        # the caller's frames are copied (and converted to fp32) into this ring: inc0 reads them in place, the residual
        # of DenBlock 1 reads them again 8 steps later (MemSkip skip1, bsvd_arch.py:378,394)
        ring("input", (n, in_ch, H, W), 10)
        exit_key = net.temp2["out3"].key
        fuse = getattr(ex, "fuse_head", None)
        fused_away = {net.temp1["inc0"].key} if (fuse and fuse(net.temp1)) else set()     # their outputs never exist (engine.head_fusable / pair_fusable)
        fusep = getattr(ex, "fuse_pair", None)
        if fusep:
            for blk in (net.temp1, net.temp2):
                for na, nb in (("inc0", "inc3"), ("out0", "out3")):
                    if fusep(blk, na, nb):
                        fused_away.add(blk[na].key)
        for blk in (net.temp1, net.temp2):
            h, w = H, W
            for name, sp in blk.items():
                ho, wo = (h - 1) // sp.stride + 1, (w - 1) // sp.stride + 1
                if sp.epilogue == 1:            # EPI_PS_ADD
                    shape = (n, 2 * ho, 2 * wo, sp.cout_pad // 4)
                    h, w = 2 * ho, 2 * wo
                else:
                    shape = (n, ho, wo, sp.cout_pad)
                    h, w = ho, wo
                if sp.key == exit_key:          # planar [n, out_ch, H, W]: what the caller gets a copy of
                    shape = (n, sp.cout, ho, wo)
                if sp.key in fused_away:
                    continue
                ring(sp.key, shape, ring_depth(name, blk is net.temp1 and name == "out3", sp.key == exit_key), out_v(sp) if out_v else 0)
        self.t1 = _DenBlockStream(net.temp1)
        self.t2 = _DenBlockStream(net.temp2)
        self.r1 = _Recorder(self)
        self.r2 = _Recorder(self)
        self.n_in = 0
        self.plans = {}
        self.graphs = OrderedDict()
        self.max_graphs = max_graphs
        self.stats = {"graph_replays": 0, "graph_captures": 0, "batch_launches": 0, "steps": 0}
        self.layerwise = False      # measurement aid: issue every plan layer by layer through ex.conv (per-launch HIP events)
        self._cap = self._side = None
        self._lag = None            # streaming_forward: DenBlock 1's output of the previous step, not yet fed to DenBlock 2
        self._mode = None

    # ---- state ---------------------------------------------------------------------------------
    def reset(self):
        """DenBlock.reset (bsvd_arch.py:352-356): only the BiBufferConv state."""
        self.t1.reset()
        self.t2.reset()

    def clear(self):
        """New stream: buffers, skip FIFOs and ring positions (so that pipeline-fill plans repeat from clip to clip)."""
        self.t1.clear()
        self.t2.clear()
        self.r1.reset()
        self.r2.reset()
        self.n_in = 0
        self._lag = None
        self._mode = None           # 'feed' | 'lagged': the two step protocols must not be mixed inside one stream

    def release(self):
        """Frees graphs and rings.  Replays may still be in flight on whatever streams the caller used (LiveStream's compute
        stream, the capture / side streams): the device is drained first, so neither a graph is destroyed under a running
        replay nor a ring block handed back to the caching allocator while a kernel on another stream still uses it."""
        if self.hip:
            torch.cuda.synchronize(self.ex.device)
        for g, _ in self.graphs.values():
            if g is not None:
                self.ex.lib.bsvd_graph_destroy(g)
        self.graphs.clear()
        self.plans.clear()
        self.rings.clear()

    # ---- planning ------------------------------------------------------------------------------
    def _stage_input(self, x):
        """x: the caller's [T,C,H,W] frame(s), T <= chunk (any float dtype / device) -> fp32 copy in the input ring."""
        if x is None:
            return None
        ring = self.rings["input"]
        slot = ring[self.n_in % len(ring)]
        self.n_in += 1
        if isinstance(x, (list, tuple)):          # a chunk handed over as the reference's list of [1,C,H,W] frames
            if len(x) != slot.shape[0]:
                slot = slot[:len(x)]
            for i, f in enumerate(x):
                slot[i:i + 1].copy_(f)
            return slot
        if x.shape[0] != slot.shape[0]:
            slot = slot[:x.shape[0]]
        slot.copy_(x)
        return slot

    def _plan(self, blk, recorder, x, x_planar, y_planar, shared_chip=False):
        """shared_chip: the plan runs as one of two concurrent graph branches (feed_lagged) -- its Winograd launches keep the full
        tile (a grid that leaves CUs idle leaves them to the other branch; measured 345 vs 330 frames/s at 540 x 960, same bits)."""
        y = blk.feed(recorder, x, x_planar=x_planar, y_planar=y_planar)
        rec = recorder.take()
        sig = _signature(rec)
        pk = getattr(self.ex, "packed", None)
        shared_chip = bool(shared_chip and self.hip and pk is not None and getattr(pk, "wino", None) and any(v in (2, 6) for v in pk.wino_layer_abi.values()))
        if shared_chip:             # (only an engine whose launches pick their own tile keeps two sets of plans)
            sig = ("shared", sig)
        plan = self.plans.get(sig)
        if plan is None:
            plan = _Plan(rec)
            if self.hip:
                plan.args = (self.ex.lib_args_type() * max(plan.n, 1))()
                for i, r in enumerate(rec):
                    plan.args[i], _ = self.ex.build_args(*r[:9], out=r[9], head=r[10], shared_chip=shared_chip, pre=r[11])
            self.plans[sig] = plan
        return y, sig, plan

    # ---- issuing -------------------------------------------------------------------------------
    def _issue_generic(self, plan):
        """plain executor (CPU tests): layer by layer, in order"""
        for r in plan.rec:
            sp, x, hp, hn, extra, eps, ecs, xpl, ypl, o = r[:10]
            if r[10] is not None:
                self.ex.conv_head_fused(r[10], sp, x, out=o)
                continue
            if r[11] is not None:
                self.ex.conv_pair_fused(r[11], sp, x, extra=extra, extra_pstride=eps, extra_cstride=ecs, y_planar=ypl, out=o)
                continue
            self.ex.conv(sp, x, halo_prev=hp, halo_next=hn, extra=extra, extra_pstride=eps, extra_cstride=ecs,
                         x_planar=xpl, y_planar=ypl, out=o)

    def _streams(self):
        if self._cap is None:
            self._cap = torch.cuda.Stream(self.ex.device)
            self._side = torch.cuda.Stream(self.ex.device)
        return ctypes.c_void_p(self._cap.cuda_stream), ctypes.c_void_p(self._side.cuda_stream)

    def _issue(self, key, plans_main, plans_side):
        """plans_main run in order; plans_side (optional) are independent of them (a parallel graph branch)."""
        if not self.hip or self.layerwise:
            for p in list(plans_side) + list(plans_main):
                self._issue_generic(p)
            return
        from . import _lib
        from .engine import _stream_ptr
        lib = self.ex.lib
        cur = _stream_ptr()
        g = self.graphs.get(key)
        if g is not None:
            self.graphs.move_to_end(key)
            if g[0] is not None:
                _lib.check(lib.bsvd_graph_launch(g[0], cur), "bsvd_graph_launch")
                self.stats["graph_replays"] += 1
                return
        n_total = sum(p.n for p in plans_main) + sum(p.n for p in plans_side)
        if self.use_graphs and g is not None and n_total > 1:
            # second sighting: capture (nothing executes), instantiate, replay
            cap, side = self._streams()
            _lib.check(lib.bsvd_graph_begin(cap), "bsvd_graph_begin")
            try:
                fork = any(p.n for p in plans_side) and any(p.n for p in plans_main)
                if fork:
                    _lib.check(lib.bsvd_graph_fork(cap, side), "bsvd_graph_fork")
                for p in plans_main:
                    if p.n:
                        _lib.check(lib.bsvd_conv3x3_batch(p.args, p.n, cap), "bsvd_conv3x3_batch (capture)")
                for p in plans_side:
                    if p.n:
                        _lib.check(lib.bsvd_conv3x3_batch(p.args, p.n, side if fork else cap), "bsvd_conv3x3_batch (capture)")
                if fork:
                    _lib.check(lib.bsvd_graph_join(cap, side), "bsvd_graph_join")
            except Exception:
                lib.bsvd_graph_abort(cap)
                raise
            exe, nn = ctypes.c_void_p(), ctypes.c_int32()
            _lib.check(lib.bsvd_graph_end(cap, ctypes.byref(exe), ctypes.byref(nn)), "bsvd_graph_end")
            self.graphs[key] = (exe, nn.value)
            self.stats["graph_captures"] += 1
            while len(self.graphs) > self.max_graphs:
                _, old = self.graphs.popitem(last=False)
                if old[0] is not None:
                    lib.bsvd_graph_destroy(old[0])
            _lib.check(lib.bsvd_graph_launch(exe, cur), "bsvd_graph_launch")
            self.stats["graph_replays"] += 1
            return
        if g is None:
            self.graphs[key] = (None, 0)          # first sighting: remember, issue directly
        for p in list(plans_main) + list(plans_side):
            if p.n:
                _lib.check(lib.bsvd_conv3x3_batch(p.args, p.n, cur), "bsvd_conv3x3_batch")
                self.stats["batch_launches"] += 1

    def _enter(self, mode):
        if self._mode is None:
            self._mode = mode
        elif self._mode != mode:
            raise RuntimeError("this stream was started with %s steps and continued with %s steps: DenBlock 2 runs in step with "
                               "DenBlock 1 in the one (feedin_one_element) and one step behind in the other (feed_overlapped / "
                               "streaming_forward); reset() between them" % (self._mode, mode))

    # ---- the two entry points ------------------------------------------------------------------
    def feed(self, x, y_planar):
        """One ``feedin_one_element`` step: both DenBlocks of this step, in order.  Returns a VIEW of the exit ring slot
        (valid until the next-but-one step; callers copy it) or None."""
        self._enter("feed")
        self.stats["steps"] += 1
        xin = self._stage_input(x)
        y1, s1, p1 = self._plan(self.t1, self.r1, xin, True, None)
        y2, s2, p2 = self._plan(self.t2, self.r2, y1, False, y_planar)
        self._issue((s1, s2), (p1, p2), ())
        return y2

    def feed_lagged(self, x, y_planar, last=False):
        """``streaming_forward`` step k: DenBlock 1 of step k and DenBlock 2 of step k-1 as two independent chains (one graph
        with two branches).  Returns DenBlock 2's result of step k-1 (a ring view; None for k = 0).  ``last``: only drain the
        lagging DenBlock-2 step (no DenBlock-1 work)."""
        self._enter("lagged")
        self.stats["steps"] += 1
        had_lag, lag = self._lag is not None, (self._lag[0] if self._lag is not None else None)
        plans_a, plans_b, sa, sb, y2 = (), (), (), (), None
        if not last:
            xin = self._stage_input(x)
            y1, sa, pa = self._plan(self.t1, self.r1, xin, True, None, shared_chip=True)
            plans_a = (pa,)
            self._lag = (y1,)
        else:
            self._lag = None
        if had_lag:
            y2, sb, pb = self._plan(self.t2, self.r2, lag, False, y_planar, shared_chip=True)
            plans_b = (pb,)
        self._issue(("lag", sa, sb), plans_b, plans_a)
        return y2
