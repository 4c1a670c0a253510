"""MIMO / segmented inference: bsvd_amd.TSN + bsvd_amd.denoise_seq + bsvd_amd.global_queue_buffer against
the golden produced by the reference's TSN + denoise_seq (tests/golden/g10_mimo_segments.npz).  The arithmetic is done
by the oracle-backed executor (the test subclass overrides the two device hooks); what is under test is the product's
host logic: segment cutting, look-ahead, mirrored tail, past-slice queue discipline, TSN checkpoint key schema."""
import numpy as np
import pytest
import torch

from helpers import load_golden, maxabs
from oracle_exec import OracleExecutor
from seeded import seeded_state, state_digest
import bsvd_amd
from bsvd_amd import global_queue_buffer as gq
from bsvd_amd.arch import TSN


class CpuTSN(TSN):
    """test double: same class, arithmetic on CPU through the oracle executor"""

    def _device(self):
        return torch.device("cpu")

    def _executor(self, device):
        return OracleExecutor({k: v.detach() for k, v in self._engine_state().items()})

    def clip_forward(self, frames, halo_fn=None):
        from bsvd_amd.schedule import bsvd_clip, planar_ok
        ex = self._executor(None)
        halo_fn.ex = ex
        pin, pout = planar_ok(ex, self.net)
        return bsvd_clip(ex, self.net, frames.float().contiguous(), halo_fn, x_planar=pin,
                         y_planar=(self.net.out_ch, self.clamp) if pout else None)


def _net(g):
    st = seeded_state([(str(k), tuple(int(v) for v in str(s).split(","))) for k, s in zip(g["tsn_keys"], g["tsn_shapes"])],
                      int(g["seed"]))
    assert state_digest(st) == str(g["digest"])
    m = CpuTSN(precision="fp32", num_segments=3, net2d_opt=dict(chns=[32, 64, 128], mid_ch=32, in_ch=4, out_ch=3, norm="none", act="relu6",
                                             interm_ch=32, blind=False))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})      # TSN schema, strict
    return m.eval()


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_segmented_inference_matches_reference(tag):
    g = load_golden("g10_mimo_segments")
    T, psz, fbl = (int(v) for v in g["cfg_" + tag])
    seq = torch.from_numpy(g["seq_" + tag])
    nm = torch.full((T, 1) + tuple(seq.shape[-2:]), 30.0 / 255.0)
    den = bsvd_amd.denoise_seq(seq, nm, psz, _net(g), future_buffer_len=fbl)
    assert maxabs(den.numpy(), g["den_" + tag]) < 1e-4
    assert gq.qsize() == 0          # cleaned like the reference's _clean()


def test_queue_module_surface():
    gq._init(2)
    assert (gq.get_batch_index(), gq.get_future_buffer_length(), gq.qsize()) == (-1, 2, 0)
    gq.put(1); gq.put(2)
    gq.set_batch_index(3)
    assert gq.get() == 1 and gq.qsize() == 1 and gq.get_batch_index() == 3
    assert gq.set_future_buffer_length(0) == 0
    gq._clean()
    assert gq.qsize() == 0


def test_tsn_rejects_unsupported_variants():
    with pytest.raises(NotImplementedError):
        TSN(precision="fp32", shift_type="TSM_toFutureOnly", net2d_opt=dict(norm="none"))
    with pytest.raises(NotImplementedError):
        TSN(precision="fp32", net2d_opt=dict(norm="in"))
    assert "TSN" in bsvd_amd.ARCH_REGISTRY or "TSN_MI355X" in bsvd_amd.ARCH_REGISTRY
