"""Host-side schedules of the BSVD forward over an abstract layer executor.

Two schedules over the SAME fused layers (netspec.ConvSpec -> one ``bsvd_conv3x3`` launch each):

* ``denblock_clip`` / ``bsvd_clip``: layer-major over a whole clip [T,H,W,C]; the temporal shift reads
  frames t-1 / t+1 of the same tensor inside the kernel, zeros (or a neighbour shard's halo) past the
  ends.  32 launches per clip.  Equivalent to the reference's streaming loop.
* ``StreamPipeline``: frame-major, the reference's own pipeline
  (/root/reference/Experimental_root/archs/bsvd_arch.py:53-114 BiBufferConv, :308-322 MemSkip,
  :374-396 DenBlock.forward, :485-488 feedin_one_element) with its None-in/None-out protocol and
  16-step latency; the "two streaming frame buffers" of every temporal-fusion conv are device tensors
  handed to the kernel as halo_prev / x / halo_next -- no concatenated copy is ever made.

The executor (``ex``) only has to provide ``conv(spec, x, halo_prev=None, halo_next=None, extra=None,
extra_pstride=0, extra_cstride=1)`` on NHWC tensors; the product uses engine.HipExecutor (HIP kernels).
"""
from collections import deque, namedtuple

# A temporal neighbour slice: element (pixel p, channel j of the slice) lives at t.flatten()[p*pstride + coff + j]
Halo = namedtuple("Halo", ["t", "pstride", "coff"])


# ------------------------------------------------------------------------------------------ clip mode
def _base_strides(x, x_planar):
    """(pixel stride, channel stride) of the residual base tensor: NHWC [T,H,W,C] or planar [T,C,H,W]."""
    if x_planar:
        return 1, x.shape[-2] * x.shape[-1]
    return x.shape[-1], 1


def planar_ok(ex, net):
    """Can the clip's planar NCHW input / output be consumed / produced directly by the edge layers?"""
    if not getattr(ex, "planar_io", False):
        return False, False
    first, last = net.temp1["inc0"], net.temp2["out3"]
    return (first.cin in (3, 4) and first.cin_pad == 16), (last.cout <= 4 and last.cout_pad == 16)


def _fused_head(ex, S):
    fn = getattr(ex, "fuse_head", None)
    return bool(fn and fn(S))


def _fused_pair(ex, S, na, nb):
    """do layers na -> nb of block S run as one launch (engine.pair_fusable)?"""
    fn = getattr(ex, "fuse_pair", None)
    return bool(fn and fn(S, na, nb))


def _inc(ex, S, x, x_planar):
    """InputCvBlock (bsvd_arch.py:194-226): fused entry (planar input), fused pair (NHWC input) or two launches"""
    if x_planar and _fused_head(ex, S):
        return ex.conv_head_fused(S["inc0"], S["inc3"], x)          # InputCvBlock in one launch (engine.head_fusable)
    if not x_planar and _fused_pair(ex, S, "inc0", "inc3"):
        return ex.conv_pair_fused(S["inc0"], S["inc3"], x)
    a = ex.conv(S["inc0"], x, x_planar=True) if x_planar else ex.conv(S["inc0"], x)
    return ex.conv(S["inc3"], a)


def _outc(ex, S, w, base, eps, ecs, y_planar):
    """OutputCvBlock + the block's residual (bsvd_arch.py:287-306, 408-414): one launch for the pair, or two"""
    kw = dict(extra=base, extra_pstride=eps, extra_cstride=ecs)
    if y_planar is not None:
        kw["y_planar"] = y_planar
    if _fused_pair(ex, S, "out0", "out3"):
        return ex.conv_pair_fused(S["out0"], S["out3"], w, **kw)
    o = ex.conv(S["out0"], w)
    return ex.conv(S["out3"], o, **kw)


def denblock_clip(ex, S, x, halo_fn=None, x_planar=False, y_planar=None):
    """One DenBlock over a clip.  x: [T,H,W,cin_pad] NHWC (or planar [T,C,H,W] with x_planar).
    halo_fn(spec, x) -> (Halo|None, Halo|None) supplies the neighbour shards' boundary slices when the clip
    is a frame-window shard.  y_planar=(channels, clamp) makes the last layer write planar NCHW."""

    def tsm(name, v):
        sp = S[name]
        if halo_fn is None:
            return ex.conv(sp, v)
        T, C = v.shape[0], v.shape[-1]
        if hasattr(halo_fn, "start") and T >= 3:
            # overlap: the exchange of the two boundary slices runs (on the communication stream) while the
            # interior frames -- whose temporal neighbours are all local -- are convolved; the two boundary
            # frames follow once the neighbours' slices have arrived.
            pending = halo_fn.start(sp, v)
            mk = getattr(ex, "empty_out", None)
            y = mk(sp, v) if mk else v.new_empty(ex.out_shape(sp, v))
            ex.conv(sp, v[1:-1], halo_prev=Halo(v[0], C, sp.fold), halo_next=Halo(v[-1], C, 0), out=y[1:-1])
            hp, hn = pending.finish()
            ex.conv(sp, v[0:1], halo_prev=hp, halo_next=Halo(v[1], C, 0), out=y[0:1])
            ex.conv(sp, v[-1:], halo_prev=Halo(v[-2], C, sp.fold), halo_next=hn, out=y[-1:])
            return y
        hp, hn = halo_fn(sp, v)
        return ex.conv(sp, v, halo_prev=hp, halo_next=hn)

    x0 = _inc(ex, S, x, x_planar)
    d = ex.conv(S["down0"], x0)
    x1 = tsm("d0c2", tsm("d0c1", d))
    d = ex.conv(S["down1"], x1)
    x2 = tsm("d1c2", tsm("d1c1", d))
    del d
    u = tsm("u2c2", tsm("u2c1", x2))
    del x2
    v = ex.conv(S["up2"], u, extra=x1, extra_pstride=x1.shape[-1])        # PixelShuffle + skip3
    del u, x1
    v = tsm("u1c2", tsm("u1c1", v))
    w = ex.conv(S["up1"], v, extra=x0, extra_pstride=x0.shape[-1])        # PixelShuffle + skip2
    del v, x0
    eps, ecs = _base_strides(x, x_planar)                                 # residual vs. the block input
    return _outc(ex, S, w, x, eps, ecs, y_planar)


def bsvd_clip(ex, net, x, halo_fn=None, x_planar=False, y_planar=None):
    """x: NHWC-padded clip, or the planar [T,C,H,W] input when x_planar; returns NHWC-padded output, or the
    planar [T,out_ch,H,W] tensor when y_planar=(out_ch, clamp)."""
    y = denblock_clip(ex, net.temp1, x, halo_fn, x_planar=x_planar)
    return denblock_clip(ex, net.temp2, y, halo_fn, y_planar=y_planar)


# ---------------------------------------------------------------------------------------- stream mode
class _TsmStage:
    """The two frame buffers around one temporal-fusion conv (BiBufferConv, bsvd_arch.py:53-114).

    feed(frame t+1) returns the layer output for frame t.  ``mid`` = pending frame (or chunk of frames), ``past`` = the
    frame (chunk) before it (its channels [fold:2fold] are what ShiftConv reads; None = zeros at stream start).
    Like the reference, ``past`` survives a flush and is only cleared by reset()."""

    def __init__(self, spec):
        self.spec = spec
        self.mid = None
        self.past = None
        self.past_valid = False

    def reset(self):
        self.mid = None
        self.past = None
        self.past_valid = False

    def feed(self, ex, nxt):
        if self.mid is None:
            self.mid = nxt
            if nxt is not None and not self.past_valid:
                self.past, self.past_valid = None, True     # zeros
            return None
        cur, sp = self.mid, self.spec
        cpad = cur.shape[-1]
        # a step may carry a chunk of several consecutive frames ([n,H,W,C], stream_plan's chunked streaming_forward):
        # inside the chunk the kernel reads frames t-1 / t+1 itself; the halos are the previous chunk's LAST frame and
        # the next chunk's FIRST frame
        hp = None if self.past is None else Halo(self.past if self.past.shape[0] == 1 else self.past[-1:], cpad, sp.fold)
        hn = None if nxt is None else Halo(nxt, cpad, 0)
        y = ex.conv(sp, cur, halo_prev=hp, halo_next=hn)
        self.past, self.mid = cur, nxt
        return y


class _SkipFifo:
    """MemSkip (bsvd_arch.py:308-322)."""

    def __init__(self):
        self.q = deque()

    def push(self, v):
        if v is not None:
            self.q.append(v)

    def pop_if(self, partner):
        return self.q.popleft() if partner is not None else None

    def __len__(self):
        return len(self.q)


class _DenBlockStream:
    def __init__(self, S):
        self.S = S
        self.stage = {n: _TsmStage(S[n]) for n in S if S[n].tsm}
        self.skip_in, self.skip_x0, self.skip_x1 = _SkipFifo(), _SkipFifo(), _SkipFifo()

    def reset(self):       # DenBlock.reset only resets the BiBufferConvs (bsvd_arch.py:352-356)
        for st in self.stage.values():
            st.reset()

    def clear(self):
        self.reset()
        for f in (self.skip_in, self.skip_x0, self.skip_x1):
            f.q.clear()

    def _pair(self, ex, a, b, v):
        return self.stage[b].feed(ex, self.stage[a].feed(ex, v))

    def feed(self, ex, x, x_planar=False, y_planar=None):
        S = self.S
        self.skip_in.push(x)
        x0 = None
        if x is not None:
            x0 = _inc(ex, S, x, x_planar)
        self.skip_x0.push(x0)
        d = None if x0 is None else ex.conv(S["down0"], x0)
        x1 = self._pair(ex, "d0c1", "d0c2", d)
        self.skip_x1.push(x1)
        d = None if x1 is None else ex.conv(S["down1"], x1)
        x2 = self._pair(ex, "d1c1", "d1c2", d)
        u = self._pair(ex, "u2c1", "u2c2", x2)
        sk = self.skip_x1.pop_if(u)
        v = None if u is None else ex.conv(S["up2"], u, extra=sk, extra_pstride=sk.shape[-1])
        v = self._pair(ex, "u1c1", "u1c2", v)
        sk = self.skip_x0.pop_if(v)
        w = None if v is None else ex.conv(S["up1"], v, extra=sk, extra_pstride=sk.shape[-1])
        base = self.skip_in.pop_if(w)
        if w is None:
            return None
        eps, ecs = _base_strides(base, x_planar)
        return _outc(ex, S, w, base, eps, ecs, y_planar)


class StreamPipeline:
    """feedin_one_element on device buffers: x is a [1,H,W,cin_pad] NHWC tensor or None (flush)."""

    def __init__(self, net):
        self.t1 = _DenBlockStream(net.temp1)
        self.t2 = _DenBlockStream(net.temp2)
        self.shift_num = net.shift_num

    def reset(self):
        self.t1.reset()
        self.t2.reset()

    def clear(self):
        self.t1.clear()
        self.t2.clear()

    def feed(self, ex, x, x_planar=False, y_planar=None):
        return self.t2.feed(ex, self.t1.feed(ex, x, x_planar=x_planar), y_planar=y_planar)
