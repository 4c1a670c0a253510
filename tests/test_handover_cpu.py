"""Which layers READ which tensors, recorded from the schedules themselves: the plain-fp32 / transformed-domain hand-over
(engine._SOLE_CONSUMER, BsvdConvArgs.y_f32 / x_f32) stores a tensor in a format only ONE layer understands, so every schedule -- clip,
frame-major stream, sharded clip with halos -- must hand that tensor (its frames, and the slices cut from it as temporal halos) to
that layer and to nobody else.  The map in engine.py is static; this test derives the readers from a recorded launch log on CPU."""
import numpy as np
import torch

from helpers import bsvd_keys
from oracle_exec import OracleExecutor
from seeded import seeded_state, seeded_clip
from bsvd_amd.engine import _SOLE_CONSUMER
from bsvd_amd.netspec import make_netspec
from bsvd_amd.schedule import StreamPipeline, bsvd_clip, Halo


class ReaderLog(OracleExecutor):
    """OracleExecutor that remembers which layer produced every tensor (by storage) and logs (reader, producer, role) per operand."""

    def __init__(self, st):
        super().__init__(st)
        self.producer = {}          # storage pointer -> (block, layer name)
        self.reads = set()          # (producer block, producer name, reader name, role)
        self.keep = []              # storages stay alive so that pointers are never reused

    def _tag(self, t, sp):
        self.producer[t.untyped_storage().data_ptr()] = sp
        self.keep.append(t)

    def _read(self, t, sp, role):
        src = self.producer.get(t.untyped_storage().data_ptr())
        if src is not None:
            self.reads.add((src.key.split(".")[0], src.name, sp.key.split(".")[0], sp.name, role))

    def halo_pack(self, frame, c0, n):
        out = super().halo_pack(frame, c0, n)
        src = self.producer.get(frame.untyped_storage().data_ptr())
        if src is not None:         # the slice travels on behalf of whoever consumes the halo: it stays the producer's tensor
            self.producer[out.untyped_storage().data_ptr()] = src
            self.keep.append(out)
        return out

    def conv(self, sp, x, halo_prev=None, halo_next=None, extra=None, extra_pstride=0, extra_cstride=1,
             x_planar=False, y_planar=None, out=None):
        self._read(x, sp, "x")
        for h in (halo_prev, halo_next):
            if h is not None:
                self._read(h.t, sp, "halo")
        if extra is not None:
            self._read(extra, sp, "extra")
        y = super().conv(sp, x, halo_prev, halo_next, extra, extra_pstride, extra_cstride, x_planar, y_planar, out)
        self._tag(y, sp)
        return y


def _net():
    chns = [32, 64, 128]
    st = seeded_state(bsvd_keys(chns, 32, 4, 3, 32), 3)
    return make_netspec(chns, 32, 4, 3, "relu6", 32), st


def _check(ex, what):
    seen = set()
    for pblk, pname, rblk, rname, role in ex.reads:
        if pname in _SOLE_CONSUMER:
            assert rblk == pblk and rname == _SOLE_CONSUMER[pname] and role in ("x", "halo"), \
                "%s: %s.%s (handed over to %s only) is read by %s.%s as %s" % (what, pblk, pname, _SOLE_CONSUMER[pname], rblk, rname, role)
            seen.add((pblk, pname))
    assert seen == {(b, n) for b in ("temp1", "temp2") for n in _SOLE_CONSUMER}, (what, sorted(seen))


def test_clip_schedule_hands_every_sole_consumer_tensor_to_that_layer_only():
    net, st = _net()
    ex = ReaderLog(st)
    x = torch.from_numpy(seeded_clip((1, 4, 4, 8, 12), 4, kind="sigma30"))[0]
    bsvd_clip(ex, net, x, x_planar=True, y_planar=(net.out_ch, None))
    _check(ex, "clip")


def test_stream_schedule_hands_every_sole_consumer_tensor_to_that_layer_only():
    net, st = _net()
    ex = ReaderLog(st)
    x = torch.from_numpy(seeded_clip((1, 5, 4, 8, 12), 4, kind="sigma30"))[0]
    pipe = StreamPipeline(net)
    for t in range(x.shape[0]):
        pipe.feed(ex, x[t:t + 1], x_planar=True, y_planar=(net.out_ch, None))
    for _ in range(pipe.shift_num + 1):
        pipe.feed(ex, None, x_planar=True, y_planar=(net.out_ch, None))
    _check(ex, "stream")


def test_sharded_clip_cuts_its_halos_from_the_consumers_own_input():
    """a shard's halo slices (engine.halo_pack of the boundary frames) are read by the temporal-fusion layer whose INPUT they were cut from"""
    net, st = _net()
    ex = ReaderLog(st)
    x = torch.from_numpy(seeded_clip((1, 3, 4, 8, 12), 4, kind="sigma30"))[0]

    def halo_fn(sp, v):          # a middle shard whose neighbours hold copies of its own boundary frames
        return Halo(ex.halo_pack(v[0], sp.fold, sp.fold), sp.fold, 0), Halo(ex.halo_pack(v[-1], 0, sp.fold), sp.fold, 0)

    bsvd_clip(ex, net, x, halo_fn, x_planar=True, y_planar=(net.out_ch, None))
    _check(ex, "sharded clip")
    assert any(role == "halo" for *_, role in ex.reads)
