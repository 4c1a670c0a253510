"""Static description of the BSVD network as a list of fused 3x3-conv layers.

Each ``ConvSpec`` is one launch of ``bsvd_conv3x3`` (include/bsvd_hip.h).  The layer list restates
``DenBlock.__init__`` / ``forward`` of the reference
(/root/reference/Experimental_root/archs/bsvd_arch.py:325-396) with activation, PixelShuffle, skip add and
residual folded into the producing conv's epilogue.

Channel padding: device activations are NHWC with C padded to a multiple of 16 (zeros), so every layer
has ``cin_pad``/``cout_pad``; ``fold`` is computed from the REAL channel count like the reference does
(``c // 8``, bsvd_arch.py:43-45).
"""
from collections import OrderedDict
from dataclasses import dataclass

EPI_PLAIN, EPI_PS_ADD, EPI_RESID = 0, 1, 2
ACTS = ("none", "relu", "relu6")


def pad16(c):
    return (int(c) + 15) // 16 * 16


@dataclass(frozen=True)
class ConvSpec:
    name: str        # short name inside the DenBlock, e.g. 'd0c1'
    key: str         # state_dict prefix, e.g. 'temp1.downc0.memconv.c1.op.conv'
    cin: int
    cout: int
    stride: int
    tsm: bool        # BiBufferConv / ShiftConv layer (temporal-shift gather)
    act: str
    epilogue: int

    @property
    def cin_pad(self):
        return pad16(self.cin)

    @property
    def cout_pad(self):
        if self.epilogue == EPI_PS_ADD:
            return 4 * pad16(self.cout // 4)
        return pad16(self.cout)

    @property
    def out_channels(self):          # real channels of the tensor this layer writes
        return self.cout // 4 if self.epilogue == EPI_PS_ADD else self.cout

    @property
    def out_channels_pad(self):
        return self.cout_pad // 4 if self.epilogue == EPI_PS_ADD else self.cout_pad

    @property
    def fold(self):
        return self.cin // 8 if self.tsm else 0

    def macs(self, h, w):
        ho, wo = (h - 1) // self.stride + 1, (w - 1) // self.stride + 1
        return self.cin * self.cout * 9 * ho * wo


def norm_key_after(conv_key):
    """state_dict prefix of the normalisation layer that follows conv ``conv_key`` in the reference when norm != 'none'
    (InputCvBlock convblock.1/.4, DownBlock convblock.1, MemCvBlock b1/b2, OutputCvBlock convblock.1; the UpBlock conv
    and the exit conv have none: bsvd_arch.py:122-130, 207-216, 237-241, 263-267, 294-298), else None."""
    stage, block, tail = conv_key.split(".", 2)
    base = "%s.%s." % (stage, block)
    if tail == "memconv.c1.op.conv":
        return base + "memconv.b1"
    if tail == "memconv.c2.op.conv":
        return base + "memconv.b2"
    if block == "inc" and tail == "convblock.0":
        return base + "convblock.1"
    if block == "inc" and tail == "convblock.3":
        return base + "convblock.4"
    if tail == "convblock.0" and (block.startswith("downc") or block == "outc"):
        return base + "convblock.1"
    return None


def denblock_specs(pre, chns, in_ch, out_ch, interm_ch, act, blind=False):
    """Ordered {name: ConvSpec} for one DenBlock with state_dict prefix ``pre`` ('temp1.'/'temp2.')."""
    c0, c1, c2 = chns
    if blind:
        in_ch = 3                       # InputCvBlock, bsvd_arch.py:205-206
    L = OrderedDict()

    def add(name, key, cin, cout, stride=1, tsm=False, a=act, epi=EPI_PLAIN):
        L[name] = ConvSpec(name, pre + key, cin, cout, stride, tsm, a, epi)

    add("inc0", "inc.convblock.0", in_ch, interm_ch)
    add("inc3", "inc.convblock.3", interm_ch, c0)
    add("down0", "downc0.convblock.0", c0, c1, stride=2)
    add("d0c1", "downc0.memconv.c1.op.conv", c1, c1, tsm=True)
    add("d0c2", "downc0.memconv.c2.op.conv", c1, c1, tsm=True)
    add("down1", "downc1.convblock.0", c1, c2, stride=2)
    add("d1c1", "downc1.memconv.c1.op.conv", c2, c2, tsm=True)
    add("d1c2", "downc1.memconv.c2.op.conv", c2, c2, tsm=True)
    add("u2c1", "upc2.memconv.c1.op.conv", c2, c2, tsm=True)
    add("u2c2", "upc2.memconv.c2.op.conv", c2, c2, tsm=True)
    add("up2", "upc2.convblock.0", c2, 4 * c1, a="none", epi=EPI_PS_ADD)
    add("u1c1", "upc1.memconv.c1.op.conv", c1, c1, tsm=True)
    add("u1c2", "upc1.memconv.c2.op.conv", c1, c1, tsm=True)
    add("up1", "upc1.convblock.0", c1, 4 * c0, a="none", epi=EPI_PS_ADD)
    add("out0", "outc.convblock.0", c0, c0)
    add("out3", "outc.convblock.3", c0, out_ch, a="none", epi=EPI_RESID)
    return L


@dataclass
class NetSpec:
    chns: tuple
    mid_ch: int
    in_ch: int
    out_ch: int
    act: str
    interm_ch: int
    blind: bool
    temp1: OrderedDict
    temp2: OrderedDict

    @property
    def layers(self):
        return list(self.temp1.values()) + list(self.temp2.values())

    @property
    def shift_num(self):               # BSVD.count_shift, bsvd_arch.py:554-560
        return sum(1 for l in self.layers if l.tsm)

    @property
    def net_in_ch(self):
        return 3 if self.blind else self.in_ch

    def macs_per_frame(self, h, w):
        """Algorithmic MACs of one frame: sum Cin*Cout*9*Hout*Wout."""
        total = 0
        for blk in (self.temp1, self.temp2):
            hh, ww = h, w
            for l in blk.values():
                total += l.macs(hh, ww)
                if l.stride == 2:
                    hh, ww = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
                if l.epilogue == EPI_PS_ADD:
                    hh, ww = 2 * hh, 2 * ww
        return total


def clip_peak_bytes(net, frames, h, w):
    """Upper bound of the device bytes schedule.bsvd_clip keeps live for a [frames,C,h,w] clip (fp32 words; the
    split-fp16 mode uses the same 4 bytes per value): per DenBlock the block input (residual base), the inc
    intermediate and x0 live together at full resolution, later x0 + the decoder tensors; both blocks' peaks do not
    overlap except for temp1's output = temp2's input.  3.5 x0 covers the live set counted in tests/test_schedule_cpu.py."""
    unit = 4 * frames * h * w
    peak = 0
    for blk in (net.temp1, net.temp2):
        cin = blk["inc0"].cin_pad if blk["inc0"].cin > 4 else blk["inc0"].cin      # planar 3/4-channel clip input
        c0 = max(blk["inc0"].cout_pad, blk["inc3"].cout_pad)
        peak = max(peak, unit * (cin + 3.5 * c0))
    return int(peak + unit * net.out_ch)


def make_netspec(chns=(32, 64, 128), mid_ch=3, in_ch=4, out_ch=3, act="relu", interm_ch=30, blind=False):
    """Defaults are the reference constructor's (bsvd_arch.py:446-447).  ``blind`` follows the WNet
    semantics (only the first stage drops the noise map, wnet_models.py:252-256); the reference's own
    BSVD(blind=True) passes it to both stages and cannot run."""
    if act not in ACTS:
        raise ValueError("act must be one of %s" % (ACTS,))
    chns = tuple(int(c) for c in chns)
    if len(chns) != 3:
        raise ValueError("chns must have three entries")
    t1 = denblock_specs("temp1.", chns, in_ch, mid_ch, interm_ch, act, blind=blind)
    t2 = denblock_specs("temp2.", chns, mid_ch, out_ch, interm_ch, act, blind=False)
    return NetSpec(chns, mid_ch, in_ch, out_ch, act, interm_ch, blind, t1, t2)
