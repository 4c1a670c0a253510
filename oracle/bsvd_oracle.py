"""CPU ORACLE for the BSVD streaming bidirectional-buffer forward -- TEST INFRASTRUCTURE ONLY.

This file is a checker, not product code.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it; nothing under ``bsvd_amd/`` does.

It restates, as plain functions over a ``{state_dict key: tensor}`` mapping, what the reference
computes in ``/root/reference/Experimental_root/archs/bsvd_arch.py``.  Two independent formulations:

* ``bsvd_clip``   -- layer-major over the whole clip: the temporal shift becomes an index shift over
  the frame axis with zero frames past both clip ends.
* ``BsvdStream``  -- frame-major pipeline with per-layer 2-frame buffers and skip FIFOs, i.e. the
  reference's own schedule (bsvd_arch.py:53-114, 308-322, 374-396, 485-552) incl. the ``None``
  propagation at stream start/end.

Pinning: both are checked against golden vectors produced by the real reference
(``tests/golden/make_golden.py`` imports it in the build container) -- see
``tests/test_oracle_golden.py``.  The reference ships no tests or golden vectors of its own for this
path, so those fixtures are the only pin.

The arithmetic is ``torch.nn.functional.conv2d`` fp32 on CPU (oneDNN), the same library call the
reference makes, so this is also the honest "port" CPU baseline timed by ``bench.py``.
An independent plain-C double-accumulating conv lives in ``oracle/conv_ref.c``.
"""
from collections import deque

import torch
import torch.nn.functional as F

__all__ = ["default_cfg", "bsvd_clip", "denblock_clip", "tsm_conv_clip", "BsvdStream", "stream_forward"]


def default_cfg(**over):
    """bsvd_c64 network section (options/test/bsvd_c64.yml:85-93)."""
    cfg = dict(chns=[64, 128, 256], mid_ch=64, in_ch=4, out_ch=3, act="relu6", interm_ch=64, blind=False)
    cfg.update(over)
    return cfg


def _act(x, kind):
    # get_act_function, bsvd_arch.py:185-192; in place like the reference's act_fn(inplace=True) (:126,130,211) --
    # every call site passes a fresh conv output, and the CPU baseline should not pay allocations the reference avoids
    if kind == "relu6":
        return x.clamp_(0.0, 6.0)
    if kind == "relu":
        return x.clamp_min_(0.0)
    if kind == "none":
        return x
    raise ValueError(kind)


def _norm_key(key):
    """state_dict prefix of the normalisation layer the reference puts right after conv `key` (norm != 'none'), or None:
    InputCvBlock convblock.1/.4 (bsvd_arch.py:207-216), DownBlock convblock.1 (:237-241), MemCvBlock b1/b2 (:122-130),
    OutputCvBlock convblock.1 (:294-298); the UpBlock conv and the last conv have none (:263-267, :298)."""
    parts = key.split(".", 2)
    if len(parts) < 3:                      # a bare conv (single-layer fixtures): no block structure, no norm layer
        return None
    blk, tail = parts[1], parts[2]
    pre = key[:len(key) - len(tail)]
    if tail in ("memconv.c1.op.conv", "memconv.c2.op.conv"):
        return pre + "memconv.b" + tail[9]
    if blk == "inc" and tail in ("convblock.0", "convblock.3"):
        return pre + "convblock.%d" % (int(tail[-1]) + 1)
    if (blk.startswith("downc") or blk == "outc") and tail == "convblock.0":
        return pre + "convblock.1"
    return None


def _conv(x, P, key, stride=1):
    y = F.conv2d(x, P[key + ".weight"], P[key + ".bias"], stride=stride, padding=1)
    nk = _norm_key(key)
    if nk is not None and nk + ".running_mean" in P:
        # eval-mode nn.BatchNorm2d (get_norm_function 'bn', bsvd_arch.py:176-183): running statistics, eps 1e-5
        y = F.batch_norm(y, P[nk + ".running_mean"], P[nk + ".running_var"], P[nk + ".weight"], P[nk + ".bias"],
                         training=False, eps=1e-5)
    return y


def _ps2(x):
    # nn.PixelShuffle(2): out[c, 2h+i, 2w+j] = in[4c+2i+j, h, w]   (bsvd_arch.py:266)
    t, c4, h, w = x.shape
    c = c4 // 4
    return x.reshape(t, c, 2, 2, h, w).permute(0, 1, 4, 2, 5, 3).reshape(t, c, 2 * h, 2 * w)


# --------------------------------------------------------------------------- clip formulation
def tsm_gather_clip(X, prev_slice=None, next_slice=None):
    """S[t][c] = X[t+1][c] (c<fold) | X[t-1][c] (fold<=c<2fold) | X[t][c]   -- ShiftConv's cat, bsvd_arch.py:48-50.

    prev_slice: channels [fold:2fold] of the frame before X[0]  (None -> zeros, stream start :94)
    next_slice: channels [0:fold]     of the frame after  X[-1] (None -> zeros, stream end   :104)
    Halo slices are what a neighbouring frame-window shard would send."""
    T, C, H, W = X.shape
    fold = C // 8
    S = X.clone()
    S[:, : 2 * fold] = 0
    if T > 1:
        S[:-1, :fold] = X[1:, :fold]
        S[1:, fold:2 * fold] = X[:-1, fold:2 * fold]
    if next_slice is not None:
        S[-1, :fold] = next_slice
    if prev_slice is not None:
        S[0, fold:2 * fold] = prev_slice
    return S


def tsm_conv_clip(X, P, key, halo=None):
    h = halo or {}
    return _conv(tsm_gather_clip(X, h.get("prev"), h.get("next")), P, key)


def denblock_clip(x, P, pre, act, halos=None, taps=None):
    """One DenBlock over a clip x[T,Cin,H,W] (bsvd_arch.py:325-414, clip restatement).

    halos: optional {tsm_layer_key: {'prev': slice, 'next': slice}} for sharded evaluation.
    taps : optional dict filled with intermediate tensors."""
    halos = halos or {}

    def tsm(v, key):
        return _act(tsm_conv_clip(v, P, pre + key, halos.get(pre + key)), act)

    def tap(name, v):
        if taps is not None:
            taps[name] = v
        return v

    a = _act(_conv(x, P, pre + "inc.convblock.0"), act)
    x0 = tap("x0", _act(_conv(a, P, pre + "inc.convblock.3"), act))
    d = _act(_conv(x0, P, pre + "downc0.convblock.0", stride=2), act)
    x1 = tap("x1", tsm(tsm(d, "downc0.memconv.c1.op.conv"), "downc0.memconv.c2.op.conv"))
    d = _act(_conv(x1, P, pre + "downc1.convblock.0", stride=2), act)
    x2 = tap("x2", tsm(tsm(d, "downc1.memconv.c1.op.conv"), "downc1.memconv.c2.op.conv"))
    u = tsm(tsm(x2, "upc2.memconv.c1.op.conv"), "upc2.memconv.c2.op.conv")
    u2 = tap("u2", _ps2(_conv(u, P, pre + "upc2.convblock.0")))
    v = x1 + u2
    v = tsm(tsm(v, "upc1.memconv.c1.op.conv"), "upc1.memconv.c2.op.conv")
    u1 = tap("u1", _ps2(_conv(v, P, pre + "upc1.convblock.0")))
    w = x0 + u1
    o = _conv(_act(_conv(w, P, pre + "outc.convblock.0"), act), P, pre + "outc.convblock.3")
    tap("o", o.clone())
    out = o.clone()
    k = min(3, out.shape[1])
    out[:, :k] = x[:, :k] - o[:, :k]          # none_minus, bsvd_arch.py:408-414 (first 3 channels only)
    return out


def bsvd_clip(x, P, cfg=None, noise_map=None, halos=None, taps=None):
    """BSVD.forward restated clip-wise.  x: [N,F,C,H,W] (N*F is one long clip, bsvd_arch.py:494-496)."""
    cfg = cfg or default_cfg()
    if noise_map is not None:
        x = torch.cat([x, noise_map], dim=2)
    N, Fr, C, H, W = x.shape
    v = x.reshape(N * Fr, C, H, W)
    t1 = {} if taps is not None else None
    y = denblock_clip(v, P, "temp1.", cfg["act"], halos, t1)
    if taps is not None:
        taps.update({"t1_" + k: val for k, val in t1.items()})
        taps["t1_out"] = y
    y = denblock_clip(y, P, "temp2.", cfg["act"], halos, None)
    return y.reshape(N, Fr, y.shape[1], H, W)


# --------------------------------------------------------------------------- streaming formulation
class _BiBuffer:
    """Two-frame buffer around one temporal-fusion conv (bsvd_arch.py:53-114).

    feed(frame t+1) -> output for frame t.  ``mid`` is the pending frame, ``past`` holds channels
    [fold:2fold] of the frame before it (zeros at stream start)."""

    def __init__(self, P, key):
        self.P, self.key = P, key
        self.mid = None
        self.past = None

    def reset(self):
        self.mid = None
        self.past = None

    def feed(self, nxt):
        if self.mid is None:
            # pipeline empty: either the first frame arrives (no output yet) or we are fully drained
            self.mid = nxt
            if nxt is not None and self.past is None:
                fold = nxt.shape[1] // 8
                self.past = torch.zeros_like(nxt[:, :fold])
            return None
        cur = self.mid
        fold = cur.shape[1] // 8
        future = nxt[:, :fold] if nxt is not None else torch.zeros_like(cur[:, :fold])
        s = torch.cat([future, self.past, cur[:, 2 * fold:]], dim=1)
        y = _conv(s, self.P, self.key)
        self.past = cur[:, fold:2 * fold]
        self.mid = nxt
        return y


class _Fifo:
    """MemSkip (bsvd_arch.py:308-322): push when a value exists, pop only when the partner exists."""

    def __init__(self):
        self.q = deque()

    def push(self, v):
        if v is not None:
            self.q.append(v)

    def pop_if(self, partner):
        return self.q.popleft() if partner is not None else None


class _DenBlockStream:
    def __init__(self, P, pre, act):
        self.P, self.pre, self.act = P, pre, act
        names = ["downc0.memconv.c1", "downc0.memconv.c2", "downc1.memconv.c1", "downc1.memconv.c2",
                 "upc2.memconv.c1", "upc2.memconv.c2", "upc1.memconv.c1", "upc1.memconv.c2"]
        self.buf = {n: _BiBuffer(P, pre + n + ".op.conv") for n in names}
        self.s_in, self.s_x0, self.s_x1 = _Fifo(), _Fifo(), _Fifo()

    def reset(self):
        for b in self.buf.values():
            b.reset()

    def _mem(self, v, stem):
        for c in ("c1", "c2"):
            v = self.buf[stem + ".memconv." + c].feed(v)
            if v is not None:
                v = _act(v, self.act)
        return v

    def feed(self, x):
        P, pre, act = self.P, self.pre, self.act
        self.s_in.push(None if x is None else x[:, :3])
        x0 = None
        if x is not None:
            x0 = _act(_conv(_act(_conv(x, P, pre + "inc.convblock.0"), act), P, pre + "inc.convblock.3"), act)
        self.s_x0.push(x0)
        d = None if x0 is None else _act(_conv(x0, P, pre + "downc0.convblock.0", 2), act)
        x1 = self._mem(d, "downc0")
        self.s_x1.push(x1)
        d = None if x1 is None else _act(_conv(x1, P, pre + "downc1.convblock.0", 2), act)
        x2 = self._mem(d, "downc1")
        u = self._mem(x2, "upc2")
        u2 = None if u is None else _ps2(_conv(u, P, pre + "upc2.convblock.0"))
        sk = self.s_x1.pop_if(u2)
        v = None if u2 is None else u2 + sk
        v = self._mem(v, "upc1")
        u1 = None if v is None else _ps2(_conv(v, P, pre + "upc1.convblock.0"))
        sk = self.s_x0.pop_if(u1)
        o = None
        if u1 is not None:
            o = _conv(_act(_conv(u1 + sk, P, pre + "outc.convblock.0"), act), P, pre + "outc.convblock.3")
        base = self.s_in.pop_if(o)
        if o is None:
            return None
        k = min(3, o.shape[1])
        o[:, :k] = base[:, :k] - o[:, :k]      # in place on the conv output, as bsvd_arch.py:408-414
        return o


class BsvdStream:
    """feedin_one_element restated (bsvd_arch.py:485-488): 16-step latency, None in / None out."""

    def __init__(self, P, cfg=None):
        cfg = cfg or default_cfg()
        self.t1 = _DenBlockStream(P, "temp1.", cfg["act"])
        self.t2 = _DenBlockStream(P, "temp2.", cfg["act"])
        self.shift_num = 16

    def reset(self):
        self.t1.reset()
        self.t2.reset()

    def feed(self, x):
        return self.t2.feed(self.t1.feed(x))


def stream_forward(x, P, cfg=None, noise_map=None, schedule=None):
    """BSVD.forward via the streaming schedule (bsvd_arch.py:490-552): F data feeds, then None feeds
    until F + shift_num results were collected; the first shift_num (all None) are dropped."""
    if noise_map is not None:
        x = torch.cat([x, noise_map], dim=2)
    N, Fr, C, H, W = x.shape
    v = x.reshape(N * Fr, C, H, W)
    T = v.shape[0]
    net = BsvdStream(P, cfg)
    outs = []
    for t in range(T):
        y = net.feed(v[t:t + 1])
        outs.append(y)
        if schedule is not None:
            schedule.append((False, y is None))
    while len(outs) < T + net.shift_num:
        y = net.feed(None)
        outs.append(y)
        if schedule is not None:
            schedule.append((True, y is None))
    # the reference issues one more (discarded) flush call before it notices it is done (:541-542)
    y = net.feed(None)
    if schedule is not None:
        schedule.append((True, y is None))
    kept = outs[net.shift_num:]
    y = torch.cat(kept, dim=0)
    return y.reshape(N, Fr, y.shape[1], H, W)


def to_torch_state(state):
    """numpy state (tests/golden/seeded.py) -> torch tensors."""
    return {k: torch.as_tensor(v) for k, v in state.items()}


def tsn_to_bsvd_keys(tsn_state):
    """Checkpoint key re-map TSN/WNet schema -> BSVD schema (bsvd_arch.py:462-474 and the block load()s)."""
    out = {}
    for k, v in tsn_state.items():
        k = k[len("module."):] if k.startswith("module.") else k
        if not k.startswith("base_model.nets_list."):
            continue
        stage, rest = k[len("base_model.nets_list."):].split(".", 1)
        pre = "temp%d." % (int(stage) + 1)
        blk, tail = rest.split(".", 1)
        if blk in ("downc0", "downc1"):
            tail = tail.replace("convblock.3.", "memconv.").replace(".net.", ".op.conv.")
        elif blk in ("upc2", "upc1"):
            if tail.startswith("convblock.0."):
                tail = tail.replace("convblock.0.", "memconv.").replace(".net.", ".op.conv.")
            else:
                tail = tail.replace("convblock.1.", "convblock.0.")
        out[pre + blk + "." + tail] = v
    return out
