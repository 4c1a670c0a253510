"""``BSVD`` -- the drop-in network class behind the reference's plug-in boundary.

Same constructor, attributes, parameter names and call protocol as
/root/reference/Experimental_root/archs/bsvd_arch.py:441-560 (``BSVD``), registered in ARCH_REGISTRY
(bsvd_arch.py:440), so ``build_network({'type': 'BSVD', ...})`` / ``DenoisingModel`` / ``profile.py`` style
drivers work unchanged -- but everything under ``forward`` runs as hand-written gfx950 kernels through
the C ABI (include/bsvd_hip.h).  The module tree only HOLDS the parameters (so ``state_dict()``,
``.to()``, ``.half()``, ``.parameters()`` behave as in the reference); it is never called layer by layer.
"""
from collections import OrderedDict

import os
import warnings

import numpy as np
import torch
import torch.nn as nn

from . import checkpoint
from .netspec import clip_peak_bytes, make_netspec, norm_key_after
from .registry import register_arch
from .schedule import StreamPipeline, bsvd_clip, planar_ok


# Any (re-)registration of a Parameter, buffer or sub-module anywhere in the process bumps this counter; the engines cache
# their parameter list against it, so ``module.weight = nn.Parameter(...)``, prune / parametrize or a swapped sub-module
# are noticed at the next forward without walking the module tree on every call.
_REGISTRATION_EPOCH = [0]


def _bump_epoch(module, *_args, **_kwargs):
    # only registrations inside an engine's own module tree count (every module of it is tagged at construction and a sub-module
    # attached later inherits the tag): other models of the process building or mutating themselves leave the engines' caches alone
    if module.__dict__.get("_bsvd_owned"):
        _REGISTRATION_EPOCH[0] += 1
        for a in _args:
            if isinstance(a, nn.Module):
                for sm in a.modules():
                    sm.__dict__["_bsvd_owned"] = True


_HOOK_HANDLES = [getattr(torch.nn.modules.module, _hook)(_bump_epoch)
                 for _hook in ("register_module_parameter_registration_hook", "register_module_buffer_registration_hook",
                               "register_module_module_registration_hook")]


def remove_registration_hooks():
    """Detaches the three process-global torch.nn registration hooks this module installs at import (an embedding application that
    never swaps parameters of a live engine can drop them; call ``model.refresh_parameters()`` by hand after such a swap then)."""
    while _HOOK_HANDLES:
        _HOOK_HANDLES.pop().remove()

# Arithmetic form of the wide stride-1 layers in the split mode (engine.wino_eligible): 'direct' (3-pass implicit GEMM), 'wino2' /
# 'wino6' (1-D Winograd F(2,3) / F(6,3) along x, conv3x3_winox.hip), 'wino26' (F(2,3) on the 128 -> 128 layers, F(6,3) on the wider
# ones) = engine.WIDE_CONV.  wide_conv='auto' takes this default.  A constructor keyword only: no environment variable changes the form
# (and with it the bits) behind a model's back; measurement variants (engine.MEASURE_WIDE_CONV) need a measurement build of the library.
# F(2,3) gains in every schedule (clip +4 %, per-frame stream +5 % over 'direct'); F(6,3) is 1-4 % faster on clips and 6 % slower per
# frame at 540 x 960.  One form per model: clip, stream and sharded schedules stay bit-identical to each other.
WIDE_CONV_DEFAULT = "wino2"
# Fused 64-channel pairs (engine.pair_fusable; BsvdConvArgs.pre_w_packed): OutputCvBlock out0 -> out3 of both DenBlocks and DenBlock 2's
# InputCvBlock inc0 -> inc3 as one launch each, the 1.33 GB tensor between them never written.  Same bits as the two launches (per-layer
# arithmetic is identical), so the knob is free to differ between models.  Built for VERDICT r04 #1 and measured SLOWER on the MI355X
# (r05: the halo recompute -- 1.375x the first conv's MFMAs -- costs more than the traffic it removes), so off by
# default; fuse_pairs=True takes it.
FUSE_PAIRS_DEFAULT = False
F16X3_WEIGHT_LIMIT = 6.0e4      # |folded weight| beyond this cannot be carried as an fp16 pair (fp16 max 65504)


class _Slots(nn.Module):
    """Plain container whose children get the given names (mirrors the reference's attribute names)."""

    def __init__(self, **children):
        super().__init__()
        for k, v in children.items():
            self.add_module(k, v)


def _conv(cin, cout, stride=1):
    # parameter holder with nn.Conv2d's names/shapes/default bias init; weight re-initialised below
    return nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=True)


def _mem(c, norm="none"):
    kids = OrderedDict(c1=_Slots(op=_Slots(conv=_conv(c, c))))
    if norm == "bn":
        kids["b1"] = nn.BatchNorm2d(c)
    kids["c2"] = _Slots(op=_Slots(conv=_conv(c, c)))
    if norm == "bn":
        kids["b2"] = nn.BatchNorm2d(c)
    return _Slots(**kids)


def _seq(norm, *items):
    """nn.Sequential-style numbering of the reference's conv blocks: items = (index, module | ('bn', channels))"""
    d = OrderedDict()
    for idx, m in items:
        if isinstance(m, tuple):
            if norm == "bn":
                d[str(idx)] = nn.BatchNorm2d(m[1])
        else:
            d[str(idx)] = m
    return nn.ModuleDict(d)


def _denblock_params(chns, in_ch, out_ch, interm_ch, blind, norm="none"):
    """Parameter holders under the reference's names (bsvd_arch.py:116-306, 325-347); norm='bn' adds the BatchNorm2d
    modules where get_norm_function puts them, so that a reference state_dict loads key for key."""
    c0, c1, c2 = chns
    if blind:
        in_ch = 3
    return _Slots(
        inc=_Slots(convblock=_seq(norm, (0, _conv(in_ch, interm_ch)), (1, ("bn", interm_ch)), (3, _conv(interm_ch, c0)),
                                  (4, ("bn", c0)))),
        downc0=_Slots(convblock=_seq(norm, (0, _conv(c0, c1, 2)), (1, ("bn", c1))), memconv=_mem(c1, norm)),
        downc1=_Slots(convblock=_seq(norm, (0, _conv(c1, c2, 2)), (1, ("bn", c2))), memconv=_mem(c2, norm)),
        upc2=_Slots(memconv=_mem(c2, norm), convblock=_seq(norm, (0, _conv(c2, 4 * c1)))),
        upc1=_Slots(memconv=_mem(c1, norm), convblock=_seq(norm, (0, _conv(c1, 4 * c0)))),
        outc=_Slots(convblock=_seq(norm, (0, _conv(c0, c0)), (1, ("bn", c0)), (3, _conv(c0, out_ch)))),
    )


class _HipNet(nn.Module):
    """Shared engine plumbing of the registered arch classes: NetSpec, precision, weight (re)packing, executor."""

    def _init_engine(self, net, precision, clamp, norm='none', wide_conv='auto', fuse_pairs='auto', f32_handover='auto', v_handover='auto'):
        if precision not in ("auto", "fp32", "f16x3"):
            raise ValueError("precision must be 'auto', 'fp32' or 'f16x3'")
        if wide_conv == "auto":
            wide_conv = WIDE_CONV_DEFAULT
        from .engine import WIDE_CONV, MEASURE_WIDE_CONV
        if wide_conv not in WIDE_CONV and wide_conv not in MEASURE_WIDE_CONV:      # (measurement names: PackedNet checks the library build)
            raise ValueError("wide_conv must be 'auto' or one of %s" % (WIDE_CONV,))
        self.wide_conv = wide_conv
        self.fuse_pairs = FUSE_PAIRS_DEFAULT if fuse_pairs == "auto" else bool(fuse_pairs)
        # tensors only Winograd-form layers read travel as plain fp32 instead of fp16 pairs (engine.F32_HANDOVER_DEFAULT)
        self.f32_handover = None if f32_handover == "auto" else bool(f32_handover)
        # ... and, between two F(6,3) layers, in the TRANSFORMED domain: the producer's epilogue applies the reader's input transform once
        # (engine.V_HANDOVER_DEFAULT)
        self.v_handover = None if v_handover == "auto" else bool(v_handover)
        if norm not in ("none", "bn"):
            raise NotImplementedError("norm=%r: 'none' (the shipped configs, options/test/bsvd_c64.yml:90) and 'bn' (the "
                                      "constructor default; eval-mode statistics folded into the packed conv weights) are "
                                      "implemented; InstanceNorm needs per-frame statistics and is not" % (norm,))
        self.net = net
        self.clamp = clamp
        self.norm = norm
        # channel counts that are not multiples of 16 ride on zero padding channels (e.g. interm_ch = 30 of the blind
        # config); a temporal-fusion layer needs whole 16-channel chunks per temporal source: fold % 16 == 0
        bad = [l.key for l in net.layers if l.tsm and l.fold % 16 and not (l.fold == 8 and l.cout_pad <= 64)]
        split_ok = not bad and net.net_in_ch in (3, 4) and net.out_ch <= 4
        self._split_ok, self._split_bad = split_ok, bad
        self._packed = None
        self._packed_sig = None
        self._exec = None
        self.precision = precision             # (property: resolves 'auto', validates, see below)

    # ``precision`` is a property: the constructor keyword AND a later ``model.precision = 'fp32'`` go through the same validation,
    # and the next forward re-packs the weights in the new mode (the pack signature holds the resolved request).  Reading it gives the
    # mode that RUNS: 'f16x3' | 'fp32' -- after a weight-range fallback of 'auto' that is 'fp32' until an in-range checkpoint is loaded.
    @property
    def precision(self):
        return self.__dict__.get("_precision_eff")

    @precision.setter
    def precision(self, precision):
        if precision not in ("auto", "fp32", "f16x3"):
            raise ValueError("precision must be 'auto', 'fp32' or 'f16x3'")
        requested = precision
        if precision == "auto":
            # the split-fp16 mode carries fp32-class accuracy (2-6e-5 vs the reference goldens, budget 1e-3) at ~3x the
            # rate: take it whenever the network's channel layout admits it (the c64 / c32-sized networks do)
            precision = "f16x3" if self._split_ok else "fp32"
        elif precision == "f16x3" and not self._split_ok:
            raise ValueError("precision='f16x3' needs temporal-fusion layers with fold %% 16 == 0 or 64 channels (fold 8), "
                             "i.e. chns[1:] = (64|128k, 128k) like the c64 and c32 networks, and <= 4 input/output "
                             "channels; offending layers: %s" % (self._split_bad[:3],))
        self.precision_requested = requested
        self.__dict__["_precision_eff"] = precision
        self._precision_init = precision       # what the request resolved to; a weight-range fallback of 'auto' to fp32 lasts one pack only

    @staticmethod
    def weight_init(m):
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, nonlinearity='relu')

    def reset_params(self):
        self._reset(self)

    @classmethod
    def _reset(cls, root):
        for m in root.modules():
            cls.weight_init(m)

    def _bsvd_state(self):
        """state_dict in BSVD key names"""
        return self.state_dict()

    def _engine_state(self):
        """conv weights / biases in BSVD key names (what netspec / PackedNet index by); norm='bn': with the eval-mode
        BatchNorm of every conv folded in."""
        st = self._bsvd_state()
        if self.norm == "bn":
            eps = {name: m.eps for name, m in self._bsvd_modules() if isinstance(m, nn.BatchNorm2d)}
            st = checkpoint.fold_batchnorm(st, [l.key for l in self.net.layers], norm_key_after, eps)
        return st

    def _bsvd_modules(self):
        """(name in BSVD key space, module) pairs -- TSN overrides the name mapping"""
        return self.named_modules()

    def _signature(self):
        """Identity + version of every parameter AND buffer (BatchNorm running statistics): any in-place update, device move or
        dtype change re-packs the weights.  Called on every forward / feed, so the tensor list is cached -- walking the module
        tree cost 0.4 ms per call, 2/3 of the host time of a graph-replayed feedin_one_element.  The cache is keyed by the
        process-wide registration epoch (torch's global parameter / buffer / module registration hooks), so replacing a
        Parameter object, pruning or swapping a sub-module invalidates it; ``_apply`` and ``load_state_dict`` drop it too."""
        cached = self.__dict__.get("_sig_tensors")
        if cached is None or cached[0] != _REGISTRATION_EPOCH[0]:
            cached = self.__dict__["_sig_tensors"] = (_REGISTRATION_EPOCH[0], list(self.parameters()) + list(self.buffers()))
        return tuple((t.data_ptr(), t._version, t.device, t.dtype) for t in cached[1])

    def extra_repr(self):
        """what `print(model)` shows besides the parameter holders (profile.py:77 prints the model)"""
        n = self.net
        return ("MI355X engine: chns=%s mid_ch=%d in_ch=%d out_ch=%d act=%s interm_ch=%d blind=%s norm=%s precision=%s "
                "(requested %s), %d fused conv layers / %d temporal-fusion, %.1f GMAC per 540x960 frame"
                % (list(n.chns), n.mid_ch, n.net_in_ch, n.out_ch, n.act, n.interm_ch, n.blind, self.norm, self.precision,
                   self.precision_requested, len(n.layers), n.shift_num, n.macs_per_frame(540, 960) / 1e9))

    def _tag_owned(self):
        """marks every module of this engine's tree for the registration hooks (see _bump_epoch)"""
        for sm in self.modules():
            sm.__dict__["_bsvd_owned"] = True

    def refresh_parameters(self):
        self.__dict__.pop("_sig_tensors", None)

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop("_sig_tensors", None)
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self.__dict__.pop("_sig_tensors", None)
        return super().load_state_dict(*args, **kwargs)

    def _executor(self, device):
        from .engine import HipExecutor, PackedNet, require_hip
        if self.norm == "bn" and self.training:
            raise RuntimeError("norm='bn' runs with the BatchNorm layers folded into the conv weights, i.e. with eval-mode "
                               "running statistics; call .eval() first (DenoisingModel.test and profile.py do: "
                               "denoising_model.py:180, profile.py:80).  Training is out of scope of this engine.")
        require_hip()
        sig = (self._signature(), str(device), self._precision_init, self.wide_conv, self.fuse_pairs, self.f32_handover, self.v_handover)
        if self._packed is None or self._packed_sig != sig:
            # every re-pack starts from the precision the constructor resolved: 'auto' that fell back to exact fp32 because ONE
            # checkpoint's folded weights left fp16's range takes the split mode again when an in-range checkpoint is loaded
            self.__dict__["_precision_eff"] = self._precision_init
            # the ring/graph engines of the stream schedule bake the packed-weight addresses into their launch plans: drop them
            # BEFORE the old pack is freed (a new executor may even reuse the old one's id())
            if getattr(self, "_stream_engs", None):
                self.release_stream_buffers()
            state = self._engine_state()
            if self.precision == "f16x3":
                # fp16 range guard of the split mode: a folded weight (norm='bn' with a tiny running_var) beyond fp16's range
                # cannot be carried as a hi+lo pair.  'auto' falls back to exact fp32, an explicit 'f16x3' refuses.
                # (Activations beyond +-65504 saturate in the split store; unbounded-ReLU networks fed [0,1] images stay
                # orders of magnitude below that)
                wmax = max(float(state[l.key + ".weight"].abs().max()) for l in self.net.layers)
                if not wmax <= F16X3_WEIGHT_LIMIT:
                    if self.precision_requested == "auto":
                        warnings.warn("bsvd_amd: max |weight| after the BatchNorm fold is %.3g, outside fp16's range: "
                                      "precision='auto' falls back to exact fp32 for this network" % wmax)
                        self.__dict__["_precision_eff"] = "fp32"
                    else:
                        raise ValueError("precision='f16x3': max |weight| after the BatchNorm fold is %.3g, outside fp16's "
                                         "range (use precision='fp32' or 'auto')" % wmax)
            self._packed = PackedNet(self.net, state, device, self.precision, self.wide_conv, fuse_pairs=self.fuse_pairs,
                                     f32_handover=self.f32_handover, v_handover=self.v_handover)
            self._packed_sig = sig
            self._exec = HipExecutor(self._packed)
            self._exec_gen = getattr(self, "_exec_gen", 0) + 1
        return self._exec

    def _device(self):
        if not torch.cuda.is_available():
            raise RuntimeError("bsvd_amd needs a HIP device (the reference hard-codes .cuda() too, "
                               "bsvd_arch.py:94,104,520); no CPU fallback exists in the product path")
        p = next(self.parameters())
        return p.device if p.is_cuda else torch.device("cuda", torch.cuda.current_device())

    def clip_forward(self, frames, halo_fn=None):
        """frames: [T,C,H,W] -> [T,out_ch,H,W], layer-major over the whole clip."""
        dev = self._device()
        with torch.no_grad(), torch.cuda.device(dev):
            ex = self._executor(dev)
            out_dtype = frames.dtype if frames.dtype in (torch.float16, torch.bfloat16) else torch.float32
            pin, pout = planar_ok(ex, self.net)
            x = frames.to(device=dev, dtype=torch.float32).contiguous()
            if not pin:
                x = ex.to_nhwc(x, self.net.temp1["inc0"].cin_pad)
            y = bsvd_clip(ex, self.net, x, halo_fn, x_planar=pin,
                          y_planar=(self.net.out_ch, self.clamp) if pout else None)
            if not pout:
                y = ex.to_nchw(y, self.net.out_ch, self.clamp)
            return y.to(out_dtype)

    def _check_input(self, input, noise_map):
        if noise_map is not None:
            input = torch.cat([input, noise_map], dim=2)
        C, H, W = input.shape[-3:]
        if input.numel() == 0:
            raise ValueError("empty clip %s (the reference fails on it too: torch.cat of no frames, "
                             "bsvd_arch.py:552)" % (tuple(input.shape),))
        if C != self.net.net_in_ch:
            raise ValueError("expected %d input channels (incl. noise map), got %d" % (self.net.net_in_ch, C))
        if H % 4 or W % 4:
            raise ValueError("H and W must be multiples of 4 (two 2x scales); DenoisingModel pads the input "
                             "(denoising_model.py:133-159), got %dx%d" % (H, W))
        return input


@register_arch
class BSVD(_HipNet):
    """Bidirectional-buffer streaming video denoiser on MI355X.

    Reference arguments (bsvd_arch.py:446-447) keep their meaning and defaults.  Engine-only keywords:
      engine_mode : 'clip' (layer-major over the whole clip, 32 launches), 'stream' (the reference's frame-major
                    pipeline with 16-step latency, O(1) memory in the clip length) or 'auto' (default: 'clip' when
                    the clip's activations fit the free HBM, else 'stream').  Same function, bit-identical results.
      clamp       : optional (lo, hi) fused into the exit kernel (callers clamp to [0,1] anyway,
                    validation_seq_infer.py:24).
      stream_overlap : streaming_forward runs DenBlock 1 of step k and DenBlock 2 of step k-1 as two parallel branches of
                    one HIP graph (default True; same results, bit for bit).
      stream_rings / stream_graphs : the stream schedule runs on preallocated ring buffers and replays each step as a
                    HIP graph (both default True; False = the allocating layer-by-layer path / batched launches).
      stream_chunk : frames per pipeline step of streaming_forward ('auto': 2 when the rings fit -- within 1 % of 8 at a quarter of the memory; 1 = the reference's
                    frame-by-frame pipeline).  feedin_one_element always runs one frame per step.
      precision   : 'f16x3' (split-fp16 3-pass MFMA with fp32 accumulation: fp32-class accuracy -- 2-6e-5 max-abs on
                    bsvd_c64, budget 1e-3 -- at ~3x the throughput; needs 64-channel or 128k-channel temporal-fusion layers:
                    fold 8 or fold % 16 == 0), 'fp32' (exact fp32 MFMA, bitwise an fmaf chain) or 'auto' (default: 'f16x3'
                    when the network admits it, else 'fp32'; ``self.precision`` holds the choice).
      norm        : 'none' or 'bn' (the reference default; eval mode only: the BatchNorm layers hold their parameters
                    under the reference's names and are folded into the packed conv weights).
    """

    def __init__(self, chns=[32, 64, 128], mid_ch=3, shift_input=False, in_ch=4, out_ch=3, norm='bn', act='relu',
                 interm_ch=30, blind=False, pretrain_ckpt='./experiments/pretrained_ckpt/bsvd-64.pth',
                 engine_mode='auto', clamp=None, precision='auto', stream_overlap=True, stream_rings=True,
                 stream_graphs=True, stream_chunk='auto', wide_conv='auto', fuse_pairs='auto', f32_handover='auto', v_handover='auto'):
        super().__init__()
        if shift_input:
            raise NotImplementedError("shift_input=True (CvBlock input stage) is not used by any BSVD config; "
                                      "the reference itself is inconsistent there (SURVEY.md §8a-16)")
        if engine_mode not in ("auto", "clip", "stream"):
            raise ValueError("engine_mode must be 'auto', 'clip' or 'stream'")
        self._init_engine(make_netspec(chns, mid_ch, in_ch, out_ch, act, interm_ch, blind), precision, clamp, norm, wide_conv, fuse_pairs, f32_handover, v_handover)
        self.engine_mode = engine_mode
        self.last_mode = None          # schedule the last forward() actually ran ('clip' | 'stream')
        self.stream_overlap = bool(stream_overlap)   # streaming_forward: temp1(step k) and temp2(step k-1) as parallel graph branches
        self.stream_rings = bool(stream_rings)       # stream schedule on preallocated rings (False: allocate per layer)
        self.stream_graphs = bool(stream_graphs)     # ... replayed as HIP graphs (False: one batched launch call per step)
        self.stream_chunk = stream_chunk             # streaming_forward: frames per pipeline step ('auto' | int)
        self._stream_engs, self._stream_key = {}, None
        # Same RNG consumption as the reference constructor (each DenBlock re-initialises itself, then BSVD does it
        # again, bsvd_arch.py:350,453): a seeded run draws the same weights AND leaves the generator in the same state,
        # so the evaluation noise that follows (ValFolderDataset) is the reference's realisation.
        self.temp1 = _denblock_params(self.net.chns, in_ch, mid_ch, interm_ch, blind, norm)
        self._reset(self.temp1)
        self.temp2 = _denblock_params(self.net.chns, mid_ch, out_ch, interm_ch, False, norm)
        self._reset(self.temp2)
        self.shift_num = self.net.shift_num
        self.reset_params()
        self._tag_owned()
        self._pipe = None
        if pretrain_ckpt is not None:
            self.load(pretrain_ckpt)

    # ---- parameters ----------------------------------------------------------------------------
    def load(self, path):
        ckpt = torch.load(path, map_location="cpu")
        print("load from %s" % path)
        state = ckpt['params'] if isinstance(ckpt, dict) and 'params' in ckpt else ckpt
        self.load_state_dict(checkpoint.to_bsvd_state(state))

    # ---- streaming protocol (bsvd_arch.py:459-461, 485-488) --------------------------------------
    def reset(self):
        """Starts a new stream.  Deliberate deviation: the reference's reset() clears only the BiBufferConv state
        (bsvd_arch.py:459-461, 352-356); after an unfinished stream its MemSkip FIFOs keep stale frames and pair them
        with every later clip, forever (golden g4d records it).  Here reset() empties the skip FIFOs too, so a reset
        stream equals a fresh one and the stream schedule can never disagree with the clip schedule."""
        if self._pipe is not None:
            self._pipe.clear()
        for eng in self._stream_engs.values():
            eng.clear()

    def release_stream_buffers(self):
        """Frees the ring buffers and HIP graphs of the stream schedule (they are kept between calls otherwise)."""
        for eng in self._stream_engs.values():
            eng.release()
        self._stream_engs, self._stream_key = {}, None
        self.__dict__.pop("_ring_oom", None)          # memory may be there again: let the next stream try its rings afresh
        self.__dict__.pop("_mode_cache", None)        # ... and 'auto' re-evaluate clip vs stream for the next clip

    @property
    def _stream_eng(self):
        """the per-frame (chunk 1) ring engine, if it exists"""
        return self._stream_engs.get(1)

    def _stream_engine(self, ex, frame_shape, chunk=1):
        """Ring/graph engine (stream_plan.StreamEngine) for frames shaped ``frame_shape`` = (C,H,W), ``chunk`` frames per
        pipeline step; None if this network's edge layers cannot take planar frames (then the allocating StreamPipeline
        runs)."""
        pin, pout = planar_ok(ex, self.net)
        if not (pin and pout and self.stream_rings):
            return None
        key = (self._exec_gen, tuple(frame_shape), None if self.clamp is None else tuple(self.clamp), self.stream_graphs)
        if self._stream_key != key:
            if len(frame_shape) != 3 or frame_shape[0] != self.net.net_in_ch:
                raise ValueError("expected frames [%d,H,W], got %s" % (self.net.net_in_ch, tuple(frame_shape)))
            self.release_stream_buffers()
            if self._pipe is not None:
                self._pipe.clear()
            self._stream_key = key
        eng = self._stream_engs.get(chunk)
        if eng is None:
            from .stream_plan import StreamEngine
            for other in [c for c in self._stream_engs if c != 1 and c != chunk]:     # keep the per-frame engine + one chunked
                self._stream_engs.pop(other).release()
            if (key, chunk) in self.__dict__.setdefault("_ring_oom", set()):
                return None                      # this configuration already failed to allocate: do not retry per frame
            try:
                eng = StreamEngine(self.net, ex, frame_shape[1], frame_shape[2], frame_shape[0], chunk=chunk,
                                   use_graphs=self.stream_graphs)
            except torch.cuda.OutOfMemoryError:
                self._ring_oom.add((key, chunk))
                # the rings did not fit after all (another tenant of the device, fragmentation): the caller retries with a
                # smaller chunk and finally takes the allocating frame-by-frame pipeline, whose footprint is smaller still
                torch.cuda.empty_cache()
                return None
            self._stream_engs[chunk] = eng
        return eng

    # 'auto' never takes more frames per step than this: at 540 x 960 a step of 8 frames holds 76.7 GB of rings for +0.9 % over a step of 2
    # (19 GB; profiles/r05_stream_modes_540x960.json) -- the smallest chunk within 1 % of the best.  An explicit stream_chunk=4 / 8 still runs.
    AUTO_CHUNK_MAX = 2

    def _pick_chunk(self, F, H, W):
        """Frames per pipeline step of streaming_forward: ``stream_chunk`` if given, else AUTO_CHUNK_MAX (2) when its rings fit half of
        the free HBM (540x960: 9.7 GB per frame of chunk; 1080p: 39 GB), else 1."""
        if self.stream_chunk != "auto":
            return max(1, min(int(self.stream_chunk), F))
        from .stream_plan import ring_bytes_estimate
        dev = self._device()
        free, _ = torch.cuda.mem_get_info(dev)
        free += torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
        held = sum(e.ring_bytes for e in self._stream_engs.values())
        n = self.AUTO_CHUNK_MAX
        while n >= 2:
            if n <= F and ring_bytes_estimate(self.net, H, W, n, getattr(getattr(self, "_packed", None), "v_out", ())) <= 0.5 * (free + held):
                return n
            n //= 2
        return 1

    def prepare_stream(self, H, W, all_flush_phases=False, dtype=torch.float32):
        """Optional warm-up before going live: drives dummy streams of HxW frames through ``feedin_one_element`` until every
        launch plan of the pipeline fill and of the steady state is a captured HIP graph, so that the first real frames pay
        neither the one-time batched launches nor the ~1 ms graph captures (a stream captures a plan at its second
        sighting).  ``all_flush_phases`` also captures the 17-step flush for every stream length modulo the ring period
        (10 dummy streams instead of one).  Returns the per-frame engine's statistics."""
        from .stream_plan import RING_PERIOD
        dev = self._device()
        frame = torch.zeros((1, self.net.net_in_ch, H, W), dtype=dtype, device=dev)
        base = 2 * self.shift_num + RING_PERIOD            # fill, then one full ring period of steady-state steps
        lengths = [base + i for i in range(RING_PERIOD)] if all_flush_phases else [base]
        for n in lengths:
            for _ in range(2):                             # first pass: plans are issued directly; second: captured
                self.reset()
                for _ in range(n):
                    self.feedin_one_element(frame)
                for _ in range(self.shift_num + 1):
                    self.feedin_one_element(None)
        self.reset()
        eng = self._stream_eng
        return None if eng is None else dict(eng.stats, graphs=sum(1 for g in eng.graphs.values() if g[0]), plans=len(eng.plans))

    def feedin_one_element(self, x):
        """x: [1,C,H,W] tensor or None (flush).  Returns the frame fed ``shift_num`` steps earlier, or None.
        Runs on preallocated rings; a step whose launch pattern has been seen before is one HIP-graph replay."""
        dev = self._device()
        with torch.no_grad(), torch.cuda.device(dev):
            ex = self._executor(dev)
            if x is not None:
                self._last_dtype = x.dtype if x.dtype in (torch.float16, torch.bfloat16) else torch.float32
            if x is not None and (x.dim() != 4 or x.shape[0] != 1):
                raise ValueError("feedin_one_element expects one frame [1,C,H,W], got %s" % (tuple(x.shape),))
            eng = self._stream_engine(ex, x.shape[1:]) if x is not None else self._stream_eng
            if eng is not None or (x is None and self._pipe is None):
                if eng is None:
                    return None                  # flush before any frame: nothing is pending (like the reference)
                y = eng.feed(x, (self.net.out_ch, self.clamp))        # a view of the exit ring: hand the caller a copy
                return None if y is None else y.to(getattr(self, "_last_dtype", torch.float32), copy=True)
            if self._pipe is None:
                self._pipe = StreamPipeline(self.net)
            xin = None
            pin, pout = planar_ok(ex, self.net)
            if x is not None:
                xin = x.to(device=dev, dtype=torch.float32).contiguous()
                if not pin:
                    xin = ex.to_nhwc(xin, self.net.temp1["inc0"].cin_pad)
            y = self._pipe.feed(ex, xin, x_planar=pin, y_planar=(self.net.out_ch, self.clamp) if pout else None)
            if y is None:
                return None
            if not pout:
                y = ex.to_nchw(y, self.net.out_ch, self.clamp)
            return y.to(getattr(self, "_last_dtype", torch.float32))

    def overlap_available(self, frame_shape):
        """True if ``feed_overlapped`` can run for frames shaped (C,H,W): the ring engine exists for this network / device state
        (stream_rings, planar edge layers, the rings fit the free HBM).  Builds and caches the engine the first feed would build."""
        dev = self._device()
        with torch.no_grad(), torch.cuda.device(dev):
            return self._stream_engine(self._executor(dev), tuple(frame_shape)) is not None

    def feed_overlapped(self, x, last=False):
        """Per-frame feed for hosts that pipeline anyway (``pipeline.LiveStream`` with depth >= 2): like ``feedin_one_element``,
        but DenBlock 2 runs ONE STEP BEHIND DenBlock 1 as the second branch of the step's HIP graph, so the single-frame
        launches of two independent layer chains share the chip (540x960: 0.95 instead of 0.89 of the clip rate).  The price
        is one feed of latency: the call returns the frame fed ``shift_num + 1`` feeds earlier (None before that).  End of
        stream: ``shift_num + 1`` feeds of None, then one call with ``last=True`` (drains the lagging DenBlock-2 step), then
        ``reset()``.  Same kernels on the same data as every other schedule: bit-identical frames.  Do not mix with
        ``feedin_one_element`` inside one stream."""
        dev = self._device()
        with torch.no_grad(), torch.cuda.device(dev):
            ex = self._executor(dev)
            if x is not None:
                if x.dim() != 4 or x.shape[0] != 1:
                    raise ValueError("feed_overlapped expects one frame [1,C,H,W], got %s" % (tuple(x.shape),))
                self._last_dtype = x.dtype if x.dtype in (torch.float16, torch.bfloat16) else torch.float32
            eng = self._stream_engine(ex, x.shape[1:]) if x is not None else self._stream_eng
            if eng is None:
                if x is None:
                    return None
                raise RuntimeError("feed_overlapped needs the ring engine (stream_rings=True, planar edge layers, enough HBM); "
                                   "use feedin_one_element")
            y = eng.feed_lagged(x, (self.net.out_ch, self.clamp), last=last)
            return None if y is None else y.to(getattr(self, "_last_dtype", torch.float32), copy=True)

    def streaming_forward(self, input_seq):
        """Pipeline-style inference over a clip (bsvd_arch.py:501-552): F data feeds, then flush feeds until
        F + shift_num results exist; the first shift_num (None) are dropped; state is reset afterwards.
        With the whole list in hand the engine runs the same pipeline on rings with ``stream_chunk`` frames per step and
        (``stream_overlap``) DenBlock 2 one step behind DenBlock 1; bit-identical to the frame-by-frame loop."""
        if isinstance(input_seq, torch.Tensor):
            input_seq = [input_seq[i:i + 1] for i in np.arange(input_seq.shape[0])]
        assert type(input_seq) == list, "convert the input into a sequence"
        if len(input_seq) > 0:
            out = self._streaming_forward_rings(input_seq)
            if out is not None:
                return out
        outs = []
        try:
            for x in input_seq:
                outs.append(self.feedin_one_element(x))
            while len(outs) < self.shift_num + len(input_seq):
                outs.append(self.feedin_one_element(None))
            self.feedin_one_element(None)      # the reference's extra, discarded flush call (:541-542)
        finally:
            self.reset()                       # also after an exception: never leave stale buffers behind
        return torch.cat(outs[self.shift_num:], dim=0)

    def _streaming_forward_rings(self, input_seq):
        """The pipeline of the loop above on stream_plan.StreamEngine: n = ``stream_chunk`` consecutive frames per pipeline
        step and, with ``stream_overlap``, temp1(step k) and temp2(step k-1) -- which do not depend on each other -- as two
        parallel branches of one HIP graph.  The single-frame launches of the quarter-resolution layers do not fill 256 CUs
        on their own; chunks and two independent chains do.  Only possible here: the per-frame API (feedin_one_element)
        must return frame k-16 before it sees frame k+1."""
        dev = self._device()
        F = len(input_seq)
        x0 = input_seq[0]
        if x0.dim() != 4 or x0.shape[0] != 1:
            raise ValueError("streaming_forward expects frames [1,C,H,W], got %s" % (tuple(x0.shape),))
        with torch.no_grad(), torch.cuda.device(dev):
            ex = self._executor(dev)
            pin, pout = planar_ok(ex, self.net)
            if not (pin and pout and self.stream_rings):
                return None
            for f in input_seq:
                if f.shape != x0.shape:          # slot.copy_() would broadcast a [1,C,1,1] or [1,1,H,W] frame silently
                    raise ValueError("streaming_forward: frames of different shapes %s / %s" % (tuple(x0.shape), tuple(f.shape)))
            n = self._pick_chunk(F, x0.shape[-2], x0.shape[-1])
            eng = self._stream_engine(ex, x0.shape[1:], n)
            while eng is None and n > 1:         # ring allocation ran out of memory: halve the chunk
                n //= 2
                eng = self._stream_engine(ex, x0.shape[1:], n)
            if eng is None:
                return None                      # -> the allocating frame-by-frame pipeline of streaming_forward
            out_dtype = x0.dtype if x0.dtype in (torch.float16, torch.bfloat16) else torch.float32
            ypl = (self.net.out_ch, self.clamp)
            out = torch.empty((F, self.net.out_ch) + tuple(x0.shape[-2:]), dtype=out_dtype, device=dev)
            chunks = [input_seq[i:i + n] for i in range(0, F, n)]
            steps = len(chunks) + self.shift_num + 1                 # incl. the reference's extra, discarded flush call
            eng.clear()
            pos = 0
            try:
                if self.stream_overlap:
                    for k in range(steps + 1):                       # step k issues temp1(k) and temp2(k-1)
                        y = eng.feed_lagged(chunks[k] if k < len(chunks) else None, ypl, last=k == steps)
                        if y is not None and pos < F:
                            out[pos:pos + y.shape[0]].copy_(y)
                            pos += y.shape[0]
                else:
                    for k in range(steps):
                        y = eng.feed(chunks[k] if k < len(chunks) else None, ypl)
                        if y is not None and pos < F:
                            out[pos:pos + y.shape[0]].copy_(y)
                            pos += y.shape[0]
                assert pos == F, (pos, F)
                return out
            finally:
                eng.clear()

    # ---- clip forward (bsvd_arch.py:490-499) -----------------------------------------------------
    def forward(self, input, noise_map=None):
        # N, F, C, H, W -> (N*F, C, H, W): like the reference, N>1 is one long clip
        input = self._check_input(input, noise_map)
        N, F, C, H, W = input.shape
        frames = input.reshape(N * F, C, H, W)
        self.last_mode = self._pick_mode(N * F, H, W)
        if self.last_mode == "clip":
            try:
                out = self.clip_forward(frames)
            except torch.cuda.OutOfMemoryError:
                if self.engine_mode != "auto":
                    raise
                # the (cached) decision met a device that has less room now: same function, O(1)-memory schedule
                self.__dict__.pop("_mode_cache", None)
                torch.cuda.empty_cache()
                self.last_mode = "stream"
        if self.last_mode == "stream":
            out = self.streaming_forward(frames)
        return out.reshape(N, F, out.shape[1], H, W)

    def _pick_mode(self, frames, H, W):
        """'auto': the clip schedule keeps up to 4.5 full-resolution 64-channel tensors of the WHOLE clip live (<= 203 GB for 85
        frames of 1080p -- fits 288 GB; 4K does not); the stream schedule holds a fixed number of frames."""
        if self.engine_mode != "auto":
            return self.engine_mode
        need = clip_peak_bytes(self.net, frames, H, W)
        # decided once per geometry while nothing else changed the picture: a clip schedule that fitted keeps fitting as long
        # as this model holds no more ring memory than it did then (the memory query costs ~50 us per call otherwise)
        held = sum(e.ring_bytes for e in self._stream_engs.values())
        cached = self.__dict__.get("_mode_cache")
        if cached is not None and cached[0] == (frames, H, W, held):
            return cached[1]
        dev = self._device()
        free, _ = torch.cuda.mem_get_info(dev)
        free += torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)     # cached by torch, reusable
        mode = "clip" if need <= 0.9 * free else "stream"
        if mode == "stream" and held and need <= 0.9 * (free + held):
            # the rings a previous streaming_forward / feedin_one_element left behind are what stands in the way: give them back
            self.release_stream_buffers()
            torch.cuda.empty_cache()
            mode, held = "clip", 0
        self.__dict__["_mode_cache"] = ((frames, H, W, held), mode)
        return mode

    def count_shift(self):
        return self.net.shift_num


def _tsn_denblock_params(chns, in_ch, out_ch, interm_ch, blind, norm="none"):
    """Parameter holders under the TSN/WNet key names (wnet_models.py:126-170; TemporalShift wraps c1/c2 as `.net`)."""
    c0, c1, c2 = chns
    if blind:
        in_ch = 3

    def cv(c):
        kids = OrderedDict(c1=_Slots(net=_conv(c, c)))
        if norm == "bn":
            kids["b1"] = nn.BatchNorm2d(c)
        kids["c2"] = _Slots(net=_conv(c, c))
        if norm == "bn":
            kids["b2"] = nn.BatchNorm2d(c)
        return _Slots(**kids)

    return _Slots(
        inc=_Slots(convblock=_seq(norm, (0, _conv(in_ch, interm_ch)), (1, ("bn", interm_ch)), (3, _conv(interm_ch, c0)),
                                  (4, ("bn", c0)))),
        downc0=_Slots(convblock=_seq(norm, (0, _conv(c0, c1, 2)), (1, ("bn", c1)), (3, cv(c1)))),
        downc1=_Slots(convblock=_seq(norm, (0, _conv(c1, c2, 2)), (1, ("bn", c2)), (3, cv(c2)))),
        upc2=_Slots(convblock=_seq(norm, (0, cv(c2)), (1, _conv(c2, 4 * c1)))),
        upc1=_Slots(convblock=_seq(norm, (0, cv(c1)), (1, _conv(c1, 4 * c0)))),
        outc=_Slots(convblock=_seq(norm, (0, _conv(c0, c0)), (1, ("bn", c0)), (3, _conv(c0, out_ch)))),
    )


class _QueueHalo:
    """halo_fn of the MIMO mode: every temporal-fusion layer takes the slice the previous segment queued as its past
    halo (segments after the first) and queues the [fold:2fold] slice of its last KEPT frame for the next segment --
    batch_shift with enable_past_buffer (temporal_shift.py:53-80).  Future halo: zeros."""

    def __init__(self, ex, enable_past_buffer):
        self.ex, self.enable = ex, enable_past_buffer

    def __call__(self, sp, v):
        from . import global_queue_buffer as gq
        from .schedule import Halo
        if not self.enable:
            return None, None
        hp = Halo(gq.get(), sp.fold, 0) if gq.get_batch_index() > 0 else None
        last_kept = v.shape[0] - 1 - gq.get_future_buffer_length()
        if last_kept < 0:          # the reference's x[-1-u] raises IndexError here too (temporal_shift.py:68); never wrap around
            raise IndexError("segment of %d frame(s) with future_buffer_len %d: no kept frame to queue as the next segment's "
                             "past slice" % (v.shape[0], gq.get_future_buffer_length()))
        gq.put(self.ex.halo_pack(v[last_kept], sp.fold, sp.fold))
        return hp, None


@register_arch
class TSN(_HipNet):
    """MIMO / segmented whole-clip inference of the same network (the training-time twin the reference evaluates the
    blind checkpoint with): /root/reference/Experimental_root/archs/tsm_arch.py:11-74 + archs_2d/wnet_models.py:233-278 +
    temporal_shift_ops/temporal_shift.py:6-80.  Inference only (eval-mode ``batch_shift`` semantics): the temporal shift
    is zero padded inside each call; across the segments of ``denoise_seq`` the past slices travel through
    ``bsvd_amd.global_queue_buffer``.  Parameters live under the TSN checkpoint key names
    (``base_model.nets_list.{0,1}...``), so ``bsvd-64.pth``-style files load with ``load_state_dict`` directly."""

    def __init__(self, num_segments=11, base_model='WNet_multistage', shift_type='TSM', shift_div=8, inplace=False,
                 net2d_opt={}, enable_past_buffer=True, clamp=None, precision='auto', wide_conv='auto', fuse_pairs='auto',
                 f32_handover='auto', v_handover='auto', **kwargs):
        super().__init__()
        if base_model != 'WNet_multistage':
            raise NotImplementedError("base_model %r" % (base_model,))
        if shift_type != 'TSM' or shift_div != 8:
            raise NotImplementedError("only shift_type='TSM' with shift_div=8 is implemented (the shipped configs)")
        o = dict(chns=[32, 64, 128], mid_ch=3, shift_input=False, stage_num=2, in_ch=4, out_ch=3, norm='bn', act='relu',
                 interm_ch=30, blind=False)
        o.update(net2d_opt)
        if o['stage_num'] != 2 or o['shift_input']:
            raise NotImplementedError("TSN on MI355X supports stage_num=2, shift_input=False")
        self.num_segments = num_segments
        self.enable_past_buffer = enable_past_buffer
        self._init_engine(make_netspec(o['chns'], o['mid_ch'], o['in_ch'], o['out_ch'], o['act'], o['interm_ch'],
                                       o['blind']), precision, clamp, o['norm'], wide_conv, fuse_pairs, f32_handover, v_handover)
        n = self.net
        stages = []
        for args_ in ((o['in_ch'], o['mid_ch'], o['blind']), (o['mid_ch'], o['out_ch'], False)):
            blk = _tsn_denblock_params(n.chns, args_[0], args_[1], o['interm_ch'], args_[2], o['norm'])
            self._reset(blk)                      # wnet_models.DenBlock re-initialises itself, then WNet does (:139,:262)
            stages.append(blk)
        self.base_model = _Slots(nets_list=nn.ModuleList(stages))
        self.reset_params()
        self._tag_owned()

    def _bsvd_state(self):
        return checkpoint.to_bsvd_state(self.state_dict())

    def _bsvd_modules(self):
        for name, m in self.named_modules():
            if isinstance(m, nn.BatchNorm2d):
                yield checkpoint.tsn_key_to_bsvd(name + ".weight")[:-len(".weight")], m

    def forward(self, input, noise_map=None):
        five_d = input.dim() == 5
        if not five_d:
            input = input[None]
            noise_map = None if noise_map is None else noise_map[None]
        input = self._check_input(input, noise_map)
        N, F, C, H, W = input.shape
        dev = self._device()
        out = self.clip_forward(input.reshape(N * F, C, H, W), _QueueHalo(self._executor(dev), self.enable_past_buffer))
        out = out.reshape(N, F, out.shape[1], H, W)
        return out if five_d else out[0]
