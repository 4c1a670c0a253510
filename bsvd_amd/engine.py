"""HIP executor: runs netspec layers on the MI355X through the C ABI (libbsvd_hip.so).

PyTorch is used only as plumbing here -- device memory (caching allocator), the current HIP stream and
tensor hand-over at the nn.Module boundary.  All arithmetic happens in the hand-written kernels.
There is deliberately no CPU path: without a HIP device / the built library this module raises.
"""
import ctypes
import os

import torch

from . import _lib
from .netspec import EPI_PLAIN, EPI_PS_ADD, EPI_RESID  # noqa: F401
from .schedule import Halo  # noqa: F401


def require_hip():
    if not torch.cuda.is_available():
        raise RuntimeError("bsvd_amd needs a HIP device (MI355X); torch.cuda.is_available() is False. "
                           "There is no CPU fallback -- the CPU restatement lives in oracle/ for tests only.")
    return _lib.load()


def _stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def head_fusable(inc0, inc3, precision):
    """Can InputCvBlock's two convs (bsvd_arch.py:207-216) run as ONE launch (BsvdConvArgs.head_w_packed)?  Split-fp16 mode,
    a 3/4-channel planar entry layer whose (padded) output is whole 32-channel pairs, and a plain stride-1 second conv with
    <= 64 output channels under the same activation.  ``BSVD_FUSE_HEAD=0`` switches the fusion off (A/B measurements)."""
    return (precision == "f16x3" and os.environ.get("BSVD_FUSE_HEAD", "1") != "0"
            and inc0.cin in (3, 4) and inc0.cin_pad == 16 and inc0.stride == 1 and not inc0.tsm and inc0.epilogue == EPI_PLAIN
            and inc0.cout_pad % 32 == 0 and inc3.cin_pad == inc0.cout_pad and inc3.cout_pad <= 64 and inc3.stride == 1
            and not inc3.tsm and inc3.epilogue == EPI_PLAIN and inc0.act == inc3.act)


def pair_fusable(a, b, precision):
    """Can two consecutive plain stride-1 convs a -> b run as ONE launch (BsvdConvArgs.pre_w_packed)?  Split-fp16 mode, both without
    temporal shift, a's (padded) output = b's input in ONE or TWO whole 32-channel pairs (the kernel carries two), b with <= 64 output channels and a PLAIN or RESID
    epilogue (the planar exit included).  The 64-channel full-resolution pairs of a DenBlock: OutputCvBlock out0 -> out3 and, where the
    block has no planar entry, InputCvBlock inc0 -> inc3 (bsvd_arch.py:194-226, 287-306)."""
    return (precision == "f16x3" and a.stride == 1 and b.stride == 1 and not a.tsm and not b.tsm and a.epilogue == EPI_PLAIN
            and b.epilogue in (EPI_PLAIN, EPI_RESID) and a.cin_pad % 16 == 0 and a.cin > 4 and a.cout_pad == b.cin_pad
            and b.cin_pad % 32 == 0 and b.cin_pad <= 64 and b.cout_pad <= 64)


# Forms of the wide layers: "direct" (3-pass implicit GEMM), "wino2" (1-D Winograd F(2,3) along x; default of arch.BSVD), "wino6" (F(6,3)),
# "wino26" (F(2,3) on the 128 -> 128 layers, F(6,3) on the wider ones).  That is all the product library contains.
WIDE_CONV = ("direct", "wino2", "wino6", "wino26")
# Measurement / test variants of the same arithmetic -- accepted only when the loaded library is a MEASUREMENT build (tools/build_measure.sh,
# selected with BSVD_HIP_LIB; bsvd_build_info() & BUILD_MEASURE): name -> (F(m,3), BsvdConvArgs.wino_m).  "wino4" F(4,3); "wino2h" / "wino6h"
# always the 8-row tile; "wino2n" one tile per workgroup whatever the grid, "wino2p" the persistent form; "wino2s" 4-wave workgroups;
# "wino2b" the all-positions-per-wave kernel (conv3x3_wino.hip).  DESIGN.md 4.1d records each of them as slower.
# (The all-positions kernel's F(4,3) instantiation -- wino_m 14, 264 B of scratch -- was never parity-tested in round 4 and fails against the
#  oracle on its first test in round 5: it is not offered.)
MEASURE_WIDE_CONV = {"wino4": (4, 4), "wino2b": (2, 12), "wino2s": (2, 22), "wino2h": (2, 32), "wino6h": (6, 36),
                     "wino2n": (2, 52), "wino2p": (2, 62)}
_FORMS = {"direct": (0, 0), "wino2": (2, 2), "wino6": (6, 6), "wino26": (6, 6)}

# fp16 range of the transformed weights U = G g (bsvd_pack_weights_wino): |U| <= max |w| x the largest |G| row sum of the form
# (wino_forms.h).  A layer whose bound leaves fp16's range stays on the direct form (PackedNet).
WINO_G_ROW_SUM = {2: 1.5, 4: 14.0 / 3.0, 6: 2048.0 / 315 + 512.0 / 105 + 128.0 / 35}
F16_PAIR_LIMIT = 6.0e4          # the same limit arch.BSVD applies to the raw weights (fp16 max 65504)


def wide_conv_known(name, lib=None):
    """Is ``name`` a form this process can run?  Product names always; measurement names only on a measurement library."""
    if name in WIDE_CONV:
        return True
    if name in MEASURE_WIDE_CONV:
        lib = lib or _lib.load()
        return bool(lib.bsvd_build_info() & _lib.BUILD_MEASURE)
    return False


# Plain-fp32 hand-over (BsvdConvArgs.y_f32 / x_f32): a tensor whose only reader is a Winograd-form layer is stored as fp32 channels instead
# of fp16 pairs -- same shape and bytes -- so that the reader's input transform starts from the value (no decode, half the load
# instructions, BT on channel pairs).  producer -> its single consumer inside a DenBlock (bsvd_arch.py:374-396); x0 / x1 (skip tensors, read
# by a direct-form layer too) and everything at full resolution stay pairs.
F32_HANDOVER_DEFAULT = True
_SOLE_CONSUMER = {"down0": "d0c1", "d0c1": "d0c2", "down1": "d1c1", "d1c1": "d1c2", "d1c2": "u2c1", "u2c1": "u2c2", "u2c2": "up2",
                  "up2": "u1c1", "u1c1": "u1c2", "u1c2": "up1"}

# Transformed-domain hand-over (BsvdConvArgs.y_v / x_v, ABI v11): where producer AND sole consumer run F(6,3), the producer's
# epilogue applies the consumer's input transform once and stores the tensor as V planes; the consumer's K loop is a 16-byte copy.  Takes
# precedence over the plain-fp32 hand-over for the pairs it covers (the others -- stride-2 and PixelShuffle producers -- keep fp32).
# OFF by default: built, bit-identical to the consumer transforming plain fp32 itself, and measured (profiles/r06b .. r06e): the consumer gains
# 4-11 % per layer (layout-dependent), the producer's epilogue (BT + 4 / 3 of the bytes, nothing to hide behind at one workgroup per CU) and the patch pass cost
# 7-16 %: the C1 clip does not move (wino6: 384-390 frames/s without, 382-388 with the hand-over; wino2, the default form: 383-384).
V_HANDOVER_DEFAULT = False
V_FORM = 6              # the form whose epilogue writes V (conv3x3_winox.hip: launch_winox_cfg)


class VT:
    """A transformed-domain activation tensor.  LOGICAL shape [T, H, W, C] (what the schedules reason about: frames, slicing, halos);
    storage ``t`` = [T, bsvd_v_frame_elems(H, W, C, m)] fp32 (V planes + edge record per frame, include/bsvd_hip.h).  Quacks like the
    torch tensors the schedules pass around as far as they look: shape, frame indexing / slicing, data_ptr, device, dtype."""
    __slots__ = ("t", "H", "W", "C", "m")

    def __init__(self, t, H, W, C, m):
        self.t, self.H, self.W, self.C, self.m = t, int(H), int(W), int(C), int(m)

    @staticmethod
    def frame_elems(H, W, C, m):
        n = _lib.load().bsvd_v_frame_elems(H, W, C, m)
        if n <= 0:
            raise ValueError("bsvd_v_frame_elems(%d, %d, %d, %d)" % (H, W, C, m))
        return int(n)

    @classmethod
    def empty(cls, T, H, W, C, m, device):
        return cls(torch.empty((T, cls.frame_elems(H, W, C, m)), dtype=torch.float32, device=device), H, W, C, m)

    @property
    def shape(self):
        return ((self.t.shape[0],) if self.t.dim() == 2 else ()) + (self.H, self.W, self.C)

    def __getitem__(self, idx):
        if self.t.dim() != 2:
            raise IndexError("a single transformed frame cannot be indexed")
        return VT(self.t[idx], self.H, self.W, self.C, self.m)

    def __len__(self):
        return self.t.shape[0]

    @property
    def frame_stride(self):
        return self.t.stride(0) if self.t.dim() == 2 else self.t.numel()

    def data_ptr(self):
        return self.t.data_ptr()

    def is_contiguous(self):
        return self.t.is_contiguous()

    def fill_(self, v):
        self.t.fill_(v)
        return self

    def numel(self):
        return self.t.numel()

    device = property(lambda self: self.t.device)
    dtype = property(lambda self: self.t.dtype)
    is_cuda = property(lambda self: self.t.is_cuda)

    def blocks(self):
        """[.., H, tiles, C / 16, block floats] view of the frame's blocks (include/bsvd_hip.h: (m + 2) x 4 x 8 units of 16 B + the edge
        line per (row, tile of 8 groups, 16-channel chunk)), without the producer's edge record behind them"""
        ntx = _lib.load().bsvd_v_groups(self.W, self.m) // 8
        blk = ((self.m + 2) * 32 + 8) * 4
        n = self.H * ntx * (self.C // 16) * blk
        v = self.t[..., :n]
        return v.reshape(*v.shape[:-1], self.H, ntx, self.C // 16, blk)


WINO_MIN_CIN = 128     # narrowest layer the Winograd form takes (engine.PackedNet(wino_min_cin=...))


def wino_eligible(sp, precision, min_cin=WINO_MIN_CIN):
    """Layers the 1-D Winograd kernels (conv3x3_winox.hip) can take: the wide stride-1 layers of the split-fp16 mode --
    the temporal-fusion convs (bsvd_arch.py:21-50) and the UpBlock convs (:257-267) at >= 128 input channels.  The choice
    depends on the LAYER only (never on the clip length or frame size), so that every schedule -- clip, stream, sharded,
    MIMO -- runs the same arithmetic per layer and stays bit-identical to the others."""
    return (precision == "f16x3" and sp.stride == 1 and sp.cin_pad >= max(128, min_cin) and sp.cout_pad % 32 == 0
            and sp.epilogue in (EPI_PLAIN, EPI_PS_ADD) and (not sp.tsm or sp.fold % 16 == 0))


class PackedNet:
    """Device-resident pre-packed weights of every layer (one-time transform of the state_dict,
    cf. BSVD.load, bsvd_arch.py:462-474): {spec.key: (w_packed, bias_packed)}."""

    def __init__(self, net, state, device, precision="fp32", wide_conv="direct", wino_min_cin=None, fuse_pairs=False, f32_handover=None,
                 v_handover=None):
        lib = require_hip()
        if not wide_conv_known(wide_conv, lib):
            if wide_conv in MEASURE_WIDE_CONV:
                raise ValueError("wide_conv=%r is a measurement variant: it exists in a -DBSVD_MEASURE build of the library only "
                                 "(tools/build_measure.sh, BSVD_HIP_LIB=build/measure/libbsvd_hip.so); the product forms are %s"
                                 % (wide_conv, WIDE_CONV))
            raise ValueError("wide_conv must be one of %s" % (WIDE_CONV,))
        self.device = device
        self.precision = precision
        self.wide_conv = wide_conv
        # F(m,3) form and the wino_m code handed to the library
        self.wino_m, self.wino_abi = _FORMS[wide_conv] if wide_conv in _FORMS else MEASURE_WIDE_CONV[wide_conv]
        self.wino_min_cin = WINO_MIN_CIN if wino_min_cin is None else int(wino_min_cin)
        self.wino = {}               # {spec.key: transformed pack} of the layers that run on the Winograd kernel
        self.wino_layer_abi = {}     # {spec.key: BsvdConvArgs.wino_m} -- "wino26": F(2,3) for the 128 -> 128 layers, F(6,3) for the wider ones
        self.wino_range_fallback = []    # eligible layers kept on the direct form because max |G g| would leave fp16's range
        self.tensors = {}
        self.order = {sp.key: i for i, sp in enumerate(net.layers)}      # position in the layer-major walk (tile_order parity)
        edge = set()
        if precision == "f16x3":
            # the entry layer (planar 3/4-channel input) runs on the fp32 VALU kernel and keeps an fp32 pack; every other
            # layer, the planar-output exit layer included, is split-packed for the MFMA kernel
            edge = {net.layers[0].key}
        with torch.cuda.device(device):
            # max |w| of every layer the Winograd form could take, in ONE host round trip (a float() per layer is a device sync per layer)
            cand = [sp for sp in net.layers if self.wino_m and wino_eligible(sp, precision, self.wino_min_cin)]
            self._wmax = {}
            if cand:
                mx = torch.stack([state[sp.key + ".weight"].detach().to(device=device, dtype=torch.float32).abs().max()
                                  if state[sp.key + ".weight"].numel() else torch.zeros((), device=device) for sp in cand]).cpu()
                self._wmax = {sp.key: float(v) for sp, v in zip(cand, mx)}
            for sp in net.layers:
                w = state[sp.key + ".weight"].detach().to(device=device, dtype=torch.float32).contiguous()
                b = state.get(sp.key + ".bias")
                if b is not None:
                    b = b.detach().to(device=device, dtype=torch.float32).contiguous()
                if tuple(w.shape) != (sp.cout, sp.cin, 3, 3):
                    raise ValueError("%s.weight has shape %s, expected %s" % (sp.key, tuple(w.shape), (sp.cout, sp.cin, 3, 3)))
                bp = torch.empty(sp.cout_pad, dtype=torch.float32, device=device)
                form = self._layer_form(sp, w)
                if form is not None:
                    m, abi = form
                    n = lib.bsvd_packed_wino_weight_elems(sp.cin_pad, sp.cout_pad, m)
                    wq = torch.empty(n, dtype=torch.float32, device=device)
                    rc = lib.bsvd_pack_weights_wino(w.data_ptr(), b.data_ptr() if b is not None else None, sp.cin, sp.cout,
                                                    sp.cin_pad, sp.cout_pad, 1 if sp.epilogue == EPI_PS_ADD else 0, m,
                                                    wq.data_ptr(), bp.data_ptr(), _stream_ptr())
                    _lib.check(rc, "bsvd_pack_weights_wino(%s)" % sp.key)
                    self.wino[sp.key] = wq
                    self.wino_layer_abi[sp.key] = abi
                    self.tensors[sp.key] = (None, bp)
                    continue
                n = lib.bsvd_packed_weight_elems(sp.cin_pad, sp.cout_pad)
                wp = torch.empty(n, dtype=torch.float32, device=device)
                dt = _lib.BSVD_F16X3 if (precision == "f16x3" and sp.key not in edge) else _lib.BSVD_F32
                rc = lib.bsvd_pack_weights(w.data_ptr(), b.data_ptr() if b is not None else None, sp.cin, sp.cout,
                                           sp.cin_pad, sp.cout_pad, 1 if sp.epilogue == EPI_PS_ADD else 0, dt,
                                           wp.data_ptr(), bp.data_ptr(), _stream_ptr())
                _lib.check(rc, "bsvd_pack_weights(%s)" % sp.key)
                self.tensors[sp.key] = (wp, bp)
            # fused network entry: the first conv's weights as the MFMA operand of the second conv's kernel, keyed by the
            # second conv (only DenBlock 1 has a planar entry layer)
            self.head = {}
            blk = getattr(net, "temp1", None)
            if blk is not None and "inc0" in blk and "inc3" in blk and head_fusable(blk["inc0"], blk["inc3"], precision):
                sp0, sp3 = blk["inc0"], blk["inc3"]
                w = state[sp0.key + ".weight"].detach().to(device=device, dtype=torch.float32).contiguous()
                b = state.get(sp0.key + ".bias")
                b = None if b is None else b.detach().to(device=device, dtype=torch.float32).contiguous()
                hw = torch.empty(lib.bsvd_packed_head_weight_bytes(sp0.cout_pad) // 4, dtype=torch.float32, device=device)
                hb = torch.empty(sp0.cout_pad, dtype=torch.float32, device=device)
                rc = lib.bsvd_pack_head_weights(w.data_ptr(), b.data_ptr() if b is not None else None, sp0.cin, sp0.cout,
                                                sp0.cout_pad, hw.data_ptr(), hb.data_ptr(), _stream_ptr())
                _lib.check(rc, "bsvd_pack_head_weights(%s)" % sp0.key)
                self.head[sp3.key] = (hw, hb, sp0)
            # plain-fp32 hand-over: producer -> consumer pairs whose consumer runs a product Winograd form and whose producer can store fp32
            # (a Winograd-form layer, or a PLAIN direct-form split layer: the stride-2 convs).  Decided from the layer forms alone.
            # ... and the transformed-domain hand-over (BsvdConvArgs.y_v / x_v) where producer and consumer both run F(6,3) and the producer is a
            # PLAIN layer: {producer key: m}, {consumer key: m}.  It takes precedence over the fp32 hand-over for those pairs.
            self.f32_out, self.f32_in = set(), set()
            self.v_out, self.v_in = {}, {}
            f32 = (F32_HANDOVER_DEFAULT if f32_handover is None else f32_handover) and precision == "f16x3"
            vh = (V_HANDOVER_DEFAULT if v_handover is None else v_handover) and precision == "f16x3"
            for blk in (getattr(net, "temp1", None), getattr(net, "temp2", None)):
                if blk is None:
                    continue
                for pn, cn in _SOLE_CONSUMER.items():
                    if pn in blk and cn in blk:
                        pr, co = blk[pn], blk[cn]
                        if pr.out_channels_pad != co.cin_pad:
                            continue
                        if vh and self.wino_layer_abi.get(co.key) == V_FORM and self.wino_layer_abi.get(pr.key) == V_FORM and \
                                pr.epilogue == EPI_PLAIN and pr.stride == 1:
                            self.v_out[pr.key] = V_FORM
                            self.v_in[co.key] = V_FORM
                        elif f32 and self.wino_layer_abi.get(co.key) in (2, 6) and \
                                (pr.key in self.wino or (pr.epilogue == EPI_PLAIN and pr.key not in edge)):
                            self.f32_out.add(pr.key)
                            self.f32_in.add(co.key)
            # fused 64-channel pairs (BsvdConvArgs.pre_w_packed), keyed by the SECOND conv: both layers keep their ordinary packs (the
            # first conv's is handed over as pre_w_packed / pre_bias), so a fused and an unfused launch read the same weights
            self.pairs = {}
            if fuse_pairs:
                for blk in (getattr(net, "temp1", None), getattr(net, "temp2", None)):
                    if blk is None:
                        continue
                    for na, nb in (("inc0", "inc3"), ("out0", "out3")):
                        if na in blk and nb in blk and blk[nb].key not in self.head and pair_fusable(blk[na], blk[nb], precision) \
                                and self.tensors[blk[na].key][0] is not None and self.tensors[blk[nb].key][0] is not None:
                            self.pairs[blk[nb].key] = blk[na]
            # The packed tensors are read from whatever stream a later forward runs on (ClipPipeline's compute stream,
            # the A/B streams of streaming_forward, a graph replay): finish the one-time pack here so no consumer can see
            # half-packed weights.  (The w/b temporaries are consumed by kernels queued on this stream.)
            torch.cuda.current_stream(device).synchronize()


    def _layer_form(self, sp, w):
        """(F(m,3), wino_m code) of a layer that runs on the Winograd kernel, else None.  A property of the LAYER and its weights only
        (never of the clip length, frame size or schedule): every schedule runs the same arithmetic per layer and stays bit-identical
        to the others."""
        if not (self.wino_m and wino_eligible(sp, self.precision, self.wino_min_cin)):
            return None
        m, abi = self.wino_m, self.wino_abi
        if self.wide_conv == "wino26" and sp.cin_pad <= 128 and sp.cout_pad <= 128:
            m = abi = 2
        # fp16 range of the TRANSFORMED weights U = G g: |U| <= max |w| x the form's largest |G| row sum (1.5 for F(2,3), 15 for F(6,3)).
        # A BN-folded layer inside the raw-weight guard (arch.F16X3_WEIGHT_LIMIT) can still leave fp16's range here: it keeps the direct form
        wmax = self._wmax[sp.key] if sp.key in self._wmax else (float(w.abs().max()) if w.numel() else 0.0)
        if not wmax * WINO_G_ROW_SUM[m] <= F16_PAIR_LIMIT:
            self.wino_range_fallback.append((sp.key, wmax, m))
            import warnings
            warnings.warn("bsvd_amd: %s: max |weight| %.3g x %.3g (F(%d,3) weight transform) leaves fp16's range; this layer runs "
                          "the direct form" % (sp.key, wmax, WINO_G_ROW_SUM[m], m))
            return None
        return m, abi


class HipExecutor:
    """conv()/to_nhwc()/to_nchw() on device tensors; every call enqueues exactly one kernel on the
    current HIP stream."""

    def __init__(self, packed):
        self.lib = require_hip()
        self.packed = packed
        self.device = packed.device
        self.split = packed.precision == "f16x3"      # NHWC activations are split16 (hi|lo fp16 pairs per chunk)
        self.dtype = _lib.BSVD_F16X3 if self.split else _lib.BSVD_F32
        self.launches = 0
        # tuning override of the direct form's fat-tile threshold (BsvdConvArgs.fat_min_wgs; 0 = library default).  Read HERE, on the
        # host side of the ABI, once per executor -- the library itself reads no environment
        self.fat_min_wgs = int(os.environ.get("BSVD_FAT_MIN_WGS", "0") or 0)
        self.force_x_f32 = self.force_y_f32 = None      # tests: override the pack's plain-fp32 hand-over decision for single layers
        self.force_y_v = None                           # tests: 0 / m -- override the pack's transformed-domain output decision
        self.record_variants = False     # profiling aid: ask the library which kernel instantiation each conv uses
        self.last_variant = None

    # -- layout at the clip boundary ------------------------------------------------------------
    def to_nhwc(self, x_nchw, c_pad):
        """[T,C,H,W] fp32 contiguous device tensor -> [T,H,W,c_pad]"""
        if self.split:
            raise NotImplementedError("precision='f16x3' enters/leaves through the planar edge layers only")
        T, C, H, W = x_nchw.shape
        x_nchw = x_nchw.contiguous()
        y = torch.empty((T, H, W, c_pad), dtype=torch.float32, device=x_nchw.device)
        rc = self.lib.bsvd_nchw_to_nhwc(x_nchw.data_ptr(), y.data_ptr(), T, C, H, W, c_pad, _lib.BSVD_F32, _stream_ptr())
        _lib.check(rc, "bsvd_nchw_to_nhwc")
        self.launches += 1
        return y

    def to_nchw(self, x_nhwc, c, clamp=None):
        if self.split:
            raise NotImplementedError("precision='f16x3' enters/leaves through the planar edge layers only")
        T, H, W, c_pad = x_nhwc.shape
        y = torch.empty((T, c, H, W), dtype=torch.float32, device=x_nhwc.device)
        lo, hi = (0.0, 0.0) if clamp is None else clamp
        rc = self.lib.bsvd_nhwc_to_nchw(x_nhwc.data_ptr(), y.data_ptr(), T, c, H, W, c_pad, _lib.BSVD_F32,
                                        0 if clamp is None else 1, lo, hi, _stream_ptr())
        _lib.check(rc, "bsvd_nhwc_to_nchw")
        self.launches += 1
        return y

    def halo_pack(self, frame, c0, n):
        """compact [H,W,n] copy of channels [c0,c0+n) of one NHWC frame [H,W,C] (message to a neighbour shard)"""
        if isinstance(frame, VT):        # transformed frame: the chunks of channels [c0, c0 + n) of every row = a transformed frame of n channels
            if c0 % 16 or n % 16:
                raise ValueError("halo_pack: a transformed tensor is cut in whole 16-channel chunks")
            if frame.t.dim() == 2:
                if len(frame) != 1:
                    raise ValueError("halo_pack: one frame")
                frame = frame[0]
            out = VT.empty(1, frame.H, frame.W, n, frame.m, frame.device)[0]
            out.blocks().copy_(frame.blocks()[:, :, c0 // 16:(c0 + n) // 16])
            self.launches += 1
            return out
        H, W, C = frame.shape[-3:]
        out = torch.empty((H, W, n), dtype=torch.float32, device=frame.device)
        # split16: whole 16-channel chunks are plain float ranges; an 8-channel half chunk (fold 8) is two pieces
        dt = _lib.BSVD_F16X3 if (self.split and (n % 16 or c0 % 16)) else _lib.BSVD_F32
        rc = self.lib.bsvd_halo_pack(frame.data_ptr(), out.data_ptr(), H * W, C, c0, n, dt, _stream_ptr())
        _lib.check(rc, "bsvd_halo_pack")
        self.launches += 1
        return out

    def halo_unpack(self, slice_, frame, c0):
        """scatter a compact [H,W,n] slice into channels [c0,c0+n) of the NHWC frame [H,W,C] (in place)"""
        if isinstance(frame, VT) or isinstance(slice_, VT):
            if not (isinstance(frame, VT) and isinstance(slice_, VT)) or c0 % 16 or slice_.C % 16:
                raise ValueError("halo_unpack: a transformed slice goes into a transformed frame, in whole 16-channel chunks")
            frame.blocks()[..., c0 // 16:(c0 + slice_.C) // 16, :].copy_(slice_.blocks())
            self.launches += 1
            return frame
        H, W, C = frame.shape[-3:]
        n = slice_.shape[-1]
        if not (frame.is_contiguous() and slice_.is_contiguous()):
            raise ValueError("halo_unpack: tensors must be contiguous")
        dt = _lib.BSVD_F16X3 if (self.split and (n % 16 or c0 % 16)) else _lib.BSVD_F32
        rc = self.lib.bsvd_halo_unpack(slice_.data_ptr(), frame.data_ptr(), H * W, C, c0, n, dt, _stream_ptr())
        _lib.check(rc, "bsvd_halo_unpack")
        self.launches += 1
        return frame

    def out_shape(self, sp, x):
        """(Logical) shape of the NHWC tensor layer ``sp`` produces from NHWC input ``x``."""
        T, H, W, _ = x.shape
        Ho, Wo = (H - 1) // sp.stride + 1, (W - 1) // sp.stride + 1
        if sp.epilogue == EPI_PS_ADD:
            return (T, 2 * Ho, 2 * Wo, sp.cout_pad // 4)
        return (T, Ho, Wo, sp.cout_pad)

    def out_v(self, sp):
        """m if layer ``sp`` writes its output in the transformed domain (PackedNet.v_out), else 0"""
        if self.force_y_v is not None:
            return self.force_y_v
        return getattr(self.packed, "v_out", {}).get(sp.key, 0)

    def empty_out(self, sp, x, frames=None):
        """Uninitialised output tensor of layer ``sp`` for input ``x`` -- a torch tensor, or a VT where the layer writes the transformed domain"""
        shape = self.out_shape(sp, x)
        if frames is not None:
            shape = (frames,) + tuple(shape[1:])
        m = self.out_v(sp)
        if m:
            return VT.empty(shape[0], shape[1], shape[2], shape[3], m, x.device)
        return torch.empty(shape, dtype=torch.float32, device=x.device)

    def empty_halo(self, v, n):
        """receive buffer for an n-channel temporal halo slice of (a frame of) ``v``"""
        if isinstance(v, VT):
            return VT.empty(1, v.H, v.W, n, v.m, v.device)[0]
        return torch.empty(tuple(v.shape[-3:-1]) + (n,), dtype=torch.float32, device=v.device)

    # -- the fused layer -------------------------------------------------------------------------
    planar_io = True      # the edge kernels read planar NCHW input / write planar NCHW output directly

    def conv(self, sp, x, halo_prev=None, halo_next=None, extra=None, extra_pstride=0, extra_cstride=1,
             x_planar=False, y_planar=None, out=None):
        """x_planar: x is the caller's planar [T,C,H,W] tensor (first layer).  y_planar=(channels, clamp|None):
        write the planar [T,channels,H,W] result directly (last layer).  out: optional preallocated (contiguous,
        e.g. a frame range of a larger tensor) destination instead of a fresh allocation."""
        a, y = self.build_args(sp, x, halo_prev, halo_next, extra, extra_pstride, extra_cstride, x_planar, y_planar, out)
        if self.record_variants:
            buf = ctypes.create_string_buffer(96)
            _lib.check(self.lib.bsvd_conv3x3_variant(ctypes.byref(a), buf, 96), "bsvd_conv3x3_variant(%s)" % sp.key)
            self.last_variant = buf.value.decode()
        rc = self.lib.bsvd_conv3x3(ctypes.byref(a), _stream_ptr())
        _lib.check(rc, "bsvd_conv3x3(%s)" % sp.key)
        self.launches += 1
        return y

    def fuse_head(self, S):
        """True if block ``S``'s entry pair inc0 -> inc3 runs as one launch (see head_fusable)"""
        return "inc3" in S and S["inc3"].key in getattr(self.packed, "head", {})

    def fuse_pair(self, S, na, nb):
        """True if block ``S``'s layers na -> nb run as one launch (see pair_fusable; PackedNet(fuse_pairs=True))"""
        return nb in S and getattr(self.packed, "pairs", {}).get(S[nb].key) is S.get(na)

    def conv_pair_fused(self, spa, spb, x, extra=None, extra_pstride=0, extra_cstride=1, y_planar=None, out=None):
        """Two plain convs in one launch: x NHWC split16 -> epilogue(act(conv(act(conv(x, spa)), spb))); the tensor between them
        never exists in HBM."""
        a, y = self.build_args(spb, x, None, None, extra, extra_pstride, extra_cstride, False, y_planar, out, pre=spa)
        if self.record_variants:
            buf = ctypes.create_string_buffer(96)
            _lib.check(self.lib.bsvd_conv3x3_variant(ctypes.byref(a), buf, 96), "bsvd_conv3x3_variant(%s)" % spb.key)
            self.last_variant = buf.value.decode()
        rc = self.lib.bsvd_conv3x3(ctypes.byref(a), _stream_ptr())
        _lib.check(rc, "bsvd_conv3x3(%s + %s)" % (spa.key, spb.key))
        self.launches += 1
        return y

    def conv_head_fused(self, sp0, sp3, x, out=None):
        """InputCvBlock in one launch: x planar [T,C,H,W] -> act(conv(act(conv(x, sp0)), sp3)) as NHWC split16; the
        intermediate tensor never exists in HBM."""
        a, y = self.build_args(sp3, x, x_planar=True, out=out, head=sp0)
        if self.record_variants:
            buf = ctypes.create_string_buffer(96)
            _lib.check(self.lib.bsvd_conv3x3_variant(ctypes.byref(a), buf, 96), "bsvd_conv3x3_variant(%s)" % sp3.key)
            self.last_variant = buf.value.decode()
        rc = self.lib.bsvd_conv3x3(ctypes.byref(a), _stream_ptr())
        _lib.check(rc, "bsvd_conv3x3(%s + %s)" % (sp0.key, sp3.key))
        self.launches += 1
        return y

    @staticmethod
    def lib_args_type():
        return _lib.BsvdConvArgs

    def build_args(self, sp, x, halo_prev=None, halo_next=None, extra=None, extra_pstride=0, extra_cstride=1,
                   x_planar=False, y_planar=None, out=None, alloc=True, head=None, shared_chip=False, pre=None):
        """Validates one fused layer and fills its ``BsvdConvArgs``; returns (args, y).  ``alloc=False`` leaves ``y`` (and
        ``args.y``) unset for the caller to supply per launch (the stream plan's exit layer).  ``shared_chip``: the launch runs
        beside another graph branch -- a Winograd layer then keeps its full tile whatever the grid (same bits; CUs its grid
        leaves idle are the other branch's)."""
        a = _lib.BsvdConvArgs()
        if not x.is_contiguous():
            raise ValueError("%s: input must be contiguous" % sp.key)
        for name, t in (("input", x), ("extra", extra), ("halo_prev", halo_prev and halo_prev.t),
                        ("halo_next", halo_next and halo_next.t), ("out", out)):
            if t is not None and (not t.is_cuda or t.dtype != torch.float32 or t.device != x.device):
                raise ValueError("%s: %s must be a float32 tensor on %s (got %s on %s)"
                                 % (sp.key, name, x.device, t.dtype, t.device))
        if x_planar and head is not None:
            T, C, H, W = x.shape
            hw, hb, sp0 = self.packed.head[sp.key]
            if sp0 is not head or C != head.cin:
                raise ValueError("%s: fused entry expects the %d-channel planar input of %s" % (sp.key, head.cin, head.key))
            a.x_planar_ch = C
            a.x_frame_stride = C * H * W
            a.head_w_packed, a.head_bias = hw.data_ptr(), hb.data_ptr()
        elif x_planar:
            T, C, H, W = x.shape
            if C != sp.cin or sp.cin_pad != 16 or sp.stride != 1 or sp.tsm:
                raise ValueError("%s: planar input needs a plain stride-1 layer with <= 4 input channels" % sp.key)
            a.x_planar_ch = C
            a.x_frame_stride = C * H * W
        elif pre is not None:
            T, H, W, cin_pad = x.shape
            if self.packed.pairs.get(sp.key) is not pre or cin_pad != pre.cin_pad:
                raise ValueError("%s: fused pair expects the %d-channel input of %s" % (sp.key, pre.cin_pad, pre.key))
            pw, pb = self.packed.tensors[pre.key]
            a.x_frame_stride = H * W * cin_pad
            a.pre_w_packed, a.pre_bias, a.pre_cin, a.pre_act = pw.data_ptr(), pb.data_ptr(), pre.cin_pad, _lib.ACT[pre.act]
        else:
            T, H, W, cin_pad = x.shape
            if cin_pad != sp.cin_pad:
                raise ValueError("%s: input has %d channels, layer expects %d" % (sp.key, cin_pad, sp.cin_pad))
            a.x_frame_stride = x.frame_stride if isinstance(x, VT) else H * W * cin_pad
        xv = x.m if isinstance(x, VT) else 0
        if xv != getattr(self.packed, "v_in", {}).get(sp.key, 0) and self.force_y_v is None:
            raise ValueError("%s: the pack expects a %s input, got a %s one" % (sp.key, "transformed-domain" if not xv else "pixel-domain",
                                                                               "transformed-domain" if xv else "pixel-domain"))
        for nm, h in (("halo_prev", halo_prev), ("halo_next", halo_next)):
            if h is not None and isinstance(h.t, VT) != bool(xv):
                raise ValueError("%s: %s and the input must live in the same domain" % (sp.key, nm))
        Ho, Wo = (H - 1) // sp.stride + 1, (W - 1) // sp.stride + 1
        if y_planar is not None:
            yc, clamp = y_planar
            if sp.cout_pad != 16 or yc != sp.cout or sp.stride != 1 or sp.tsm or sp.epilogue == EPI_PS_ADD:
                raise ValueError("%s: planar output needs a plain stride-1 layer with <= 4 output channels" % sp.key)
            yshape = (T, yc, H, W)
            a.y_planar_ch = yc
            if clamp is not None:
                a.y_clamp, a.y_lo, a.y_hi = 1, float(clamp[0]), float(clamp[1])
        elif sp.epilogue == EPI_PS_ADD:
            yshape = (T, 2 * Ho, 2 * Wo, sp.cout_pad // 4)
        else:
            yshape = (T, Ho, Wo, sp.cout_pad)
        yv = self.out_v(sp) if (y_planar is None and pre is None and head is None) else 0
        if out is not None:
            if tuple(out.shape) != yshape or not out.is_contiguous() or out.dtype != torch.float32 or (out.m if isinstance(out, VT) else 0) != yv:
                raise ValueError("%s: out has shape %s, expected contiguous %s%s" % (sp.key, tuple(out.shape), yshape, " in the transformed domain" if yv else ""))
            y = out
        elif alloc:
            y = VT.empty(yshape[0], yshape[1], yshape[2], yshape[3], yv, x.device) if yv else torch.empty(yshape, dtype=torch.float32, device=x.device)
        else:
            y = None
            if yv:
                raise ValueError("%s: a transformed-domain output needs its tensor" % sp.key)
        wp, bp = self.packed.tensors[sp.key]
        a.x = x.data_ptr()
        if sp.tsm:
            a.fold = sp.fold
            if halo_prev is not None:
                a.halo_prev, a.halo_prev_pstride, a.halo_prev_coff = halo_prev.t.data_ptr(), halo_prev.pstride, halo_prev.coff
            if halo_next is not None:
                a.halo_next, a.halo_next_pstride, a.halo_next_coff = halo_next.t.data_ptr(), halo_next.pstride, halo_next.coff
        a.bias_packed = bp.data_ptr()
        yf = self.force_y_f32 if self.force_y_f32 is not None else sp.key in getattr(self.packed, "f32_out", ())
        xf = self.force_x_f32 if self.force_x_f32 is not None else sp.key in getattr(self.packed, "f32_in", ())
        if yf and not yv:
            a.y_f32 = 1
        if xf and wp is None and not xv:
            a.x_f32 = 1
        a.x_v, a.y_v = xv, yv
        if wp is not None:
            a.w_packed = wp.data_ptr()
        else:
            # the Winograd kernel has no generic gather: validate what it needs HERE, before anything is issued or captured
            # (the library would answer -19 in the middle of a forward)
            for nm, h in (("halo_prev", halo_prev), ("halo_next", halo_next)):
                if sp.tsm and h is not None:
                    if h.t.data_ptr() % 16 or h.pstride % 4 or h.coff % 4:
                        raise ValueError("%s: the Winograd form needs a 16-byte aligned %s (pointer %% 16, pstride %% 4, coff %% 4 elements); "
                                         "got pstride %d, coff %d" % (sp.key, nm, h.pstride, h.coff))
                    span = (H * (self.lib.bsvd_v_groups(W, xv) // 8) * (((xv + 2) * 32 + 8) // 16) if xv else H * W) * h.pstride * 4
                    if span >= 2 ** 31 - 1:
                        raise ValueError("%s: %s spans %d bytes >= 2 GiB (32-bit byte offsets inside one frame)" % (sp.key, nm, span))
            if x.data_ptr() % 16:
                raise ValueError("%s: the Winograd form needs a 16-byte aligned input" % sp.key)
            a.w_wino_packed, a.wino_m = self.packed.wino[sp.key].data_ptr(), self.packed.wino_layer_abi[sp.key]
            if shared_chip and a.wino_m in (2, 6):
                a.wino_m += 40        # never the half-height tile
        if extra is not None:
            a.extra = extra.data_ptr()
            a.extra_frame_stride = extra[0].numel()
            a.extra_pstride, a.extra_cstride = extra_pstride, extra_cstride
            if self.split and y_planar is not None and extra_pstride != 1:
                a.extra_split = 1           # the base of the last layer's residual is an engine (split16) tensor
        elif sp.epilogue == EPI_RESID:
            raise ValueError("%s: the residual layer needs its base tensor" % sp.key)
        a.resid_ch = min(3, sp.cout) if sp.epilogue == EPI_RESID else 0
        if y is not None:
            a.y = y.data_ptr()
        a.y_frame_stride = 1
        for d in yshape[1:]:
            a.y_frame_stride *= d
        if yv:
            a.y_frame_stride = y.frame_stride
        a.frames, a.H, a.W = T, H, W
        a.Cin, a.Cout = sp.cin_pad, sp.cout_pad
        a.stride = sp.stride
        a.act, a.epilogue, a.dtype = _lib.ACT[sp.act], sp.epilogue, self.dtype
        # consecutive layers walk their tiles in opposite directions: each starts where its producer finished (Infinity Cache)
        a.tile_order = self.packed.order.get(sp.key, 0) & 1
        a.fat_min_wgs = self.fat_min_wgs
        return a, y
