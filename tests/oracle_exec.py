"""Oracle-backed layer executor for TESTS ONLY.

Implements the executor interface of bsvd_amd.schedule (conv / to_nhwc / to_nchw / halo_pack) on CPU
tensors with the oracle's arithmetic (torch conv2d fp32, oracle/bsvd_oracle.py conventions), from the
UNPACKED state_dict.  It lets the CPU test-suite check the product's host logic (layer list, channel
padding, schedules, halo bookkeeping) against the reference goldens, and serves as the per-layer
comparator for the GPU parity tests.  Never imported by bsvd_amd.
"""
import torch
import torch.nn.functional as F

from bsvd_amd.netspec import EPI_PS_ADD, EPI_RESID


def _slice_from_halo(halo, hw, n):
    """[HW, n] view of a Halo(t, pstride, coff)."""
    flat = halo.t.reshape(-1)
    return torch.as_strided(flat, (hw, n), (halo.pstride, 1), storage_offset=flat.storage_offset() + halo.coff)


class OracleExecutor:
    def __init__(self, state, double=False, fuse_pairs=False):
        self.state = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in state.items()}
        self.double = double
        self.launches = 0
        self.log = []
        self.fuse_pairs = fuse_pairs      # claim engine.pair_fusable pairs as one launch (host-logic tests of the fused-pair plumbing)

    def fuse_pair(self, S, na, nb):
        from bsvd_amd.engine import pair_fusable
        return bool(self.fuse_pairs and na in S and nb in S and pair_fusable(S[na], S[nb], "f16x3"))

    def conv_pair_fused(self, spa, spb, x, extra=None, extra_pstride=0, extra_cstride=1, y_planar=None, out=None):
        """two plain convs as ONE executor call (what BsvdConvArgs.pre_w_packed does on the device)"""
        n0 = self.launches
        y = self.conv(spb, self.conv(spa, x), extra=extra, extra_pstride=extra_pstride, extra_cstride=extra_cstride, y_planar=y_planar)
        self.launches = n0 + 1
        self.log[-2:] = [spa.key + "+" + spb.key]
        if out is not None:
            out.copy_(y)
            return out
        return y

    def to_nhwc(self, x_nchw, c_pad):
        T, C, H, W = x_nchw.shape
        y = torch.zeros((T, H, W, c_pad), dtype=torch.float32)
        y[..., :C] = x_nchw.permute(0, 2, 3, 1)
        return y

    def to_nchw(self, x_nhwc, c, clamp=None):
        y = x_nhwc[..., :c].permute(0, 3, 1, 2).contiguous()
        if clamp is not None:
            y = y.clamp(clamp[0], clamp[1])
        return y

    def out_shape(self, sp, x):
        T, H, W, _ = x.shape
        Ho, Wo = (H - 1) // sp.stride + 1, (W - 1) // sp.stride + 1
        if sp.epilogue == EPI_PS_ADD:
            return (T, 2 * Ho, 2 * Wo, sp.cout_pad // 4)
        return (T, Ho, Wo, sp.cout_pad)

    def halo_pack(self, frame, c0, n):
        return frame[..., c0:c0 + n].contiguous()

    planar_io = True

    def conv(self, sp, x, halo_prev=None, halo_next=None, extra=None, extra_pstride=0, extra_cstride=1,
             x_planar=False, y_planar=None, out=None):
        if out is not None:
            out.copy_(self.conv(sp, x, halo_prev, halo_next, extra, extra_pstride, extra_cstride, x_planar, y_planar))
            return out
        self.launches += 1
        self.log.append(sp.key)
        if x_planar:
            T, C, H, W = x.shape
            assert C == sp.cin and sp.cin_pad == 16
            v = x.contiguous()
        else:
            T, H, W, cp = x.shape
            assert cp == sp.cin_pad, (sp.key, cp, sp.cin_pad)
            assert float(x[..., sp.cin:].abs().max()) == 0.0 if cp > sp.cin else True, "padded input channels must be zero"
            v = x[..., :sp.cin].permute(0, 3, 1, 2).contiguous()       # [T,cin,H,W]
        if sp.tsm:
            fold = sp.fold
            g = v.clone()
            g[:, :2 * fold] = 0
            if T > 1:
                g[:-1, :fold] = v[1:, :fold]
                g[1:, fold:2 * fold] = v[:-1, fold:2 * fold]
            if halo_next is not None:
                g[-1, :fold] = _slice_from_halo(halo_next, H * W, fold).t().reshape(fold, H, W)
            if halo_prev is not None:
                g[0, fold:2 * fold] = _slice_from_halo(halo_prev, H * W, fold).t().reshape(fold, H, W)
            v = g
        w, b = self.state[sp.key + ".weight"], self.state[sp.key + ".bias"]
        if self.double:
            y = F.conv2d(v.double(), w.double(), b.double(), stride=sp.stride, padding=1).float()
        else:
            y = F.conv2d(v, w, b, stride=sp.stride, padding=1)
        if sp.act == "relu6":
            y = y.clamp(0.0, 6.0)
        elif sp.act == "relu":
            y = y.clamp_min(0.0)
        Ho, Wo = y.shape[-2:]
        if sp.epilogue == EPI_PS_ADD:
            cq = sp.cout // 4
            y = y.reshape(T, cq, 2, 2, Ho, Wo).permute(0, 1, 4, 2, 5, 3).reshape(T, cq, 2 * Ho, 2 * Wo)
            out = torch.zeros((T, 2 * Ho, 2 * Wo, sp.cout_pad // 4), dtype=torch.float32)
            out[..., :cq] = y.permute(0, 2, 3, 1)
            if extra is not None:
                ef = extra.reshape(-1)
                e = torch.as_strided(ef, (T, 4 * Ho * Wo, cq), (extra[0].numel(), extra_pstride, extra_cstride),
                                     storage_offset=ef.storage_offset())
                out[..., :cq] += e.reshape(T, 2 * Ho, 2 * Wo, cq)
            return out
        out = torch.zeros((T, Ho, Wo, sp.cout_pad), dtype=torch.float32)
        out[..., :sp.cout] = y.permute(0, 2, 3, 1)
        if sp.epilogue == EPI_RESID:
            k = min(3, sp.cout)
            ef = extra.reshape(-1)
            e = torch.as_strided(ef, (T, Ho * Wo, k), (extra[0].numel(), extra_pstride, extra_cstride),
                                 storage_offset=ef.storage_offset())
            out[..., :k] = e.reshape(T, Ho, Wo, k) - out[..., :k]
        if y_planar is not None:
            yc, clamp = y_planar
            assert yc == sp.cout
            out = out[..., :yc].permute(0, 3, 1, 2).contiguous()
            if clamp is not None:
                out = out.clamp(clamp[0], clamp[1])
        return out
