"""The C-ABI shared library loads and exports every symbol include/bsvd_hip.h declares; argument
validation works without touching a device (CPU-safe: no kernel is launched)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "bsvd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bsvd_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from bsvd_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 8
    for s in syms:
        assert hasattr(lib, s), "libbsvd_hip.so does not export %s" % s
    assert sorted(_lib.EXPORTS) == syms


def test_struct_layout_matches_header():
    from bsvd_amd import _lib
    lib = _lib.load()
    assert lib.bsvd_abi_version() == _lib.ABI_VERSION
    lib.bsvd_conv_args_size.restype = ctypes.c_int
    assert lib.bsvd_conv_args_size() == ctypes.sizeof(_lib.BsvdConvArgs)


def test_argument_validation_without_device():
    from bsvd_amd import _lib
    lib = _lib.load()
    assert lib.bsvd_conv3x3(None, None) == -1
    a = _lib.BsvdConvArgs()
    a.x = a.y = a.w_packed = 16
    a.frames, a.H, a.W, a.Cin, a.Cout, a.stride = 1, 8, 8, 12, 16, 1
    assert lib.bsvd_conv3x3(ctypes.byref(a), None) == -5 and b"multiples of 16" in lib.bsvd_last_error()
    a.Cin, a.stride = 16, 3
    assert lib.bsvd_conv3x3(ctypes.byref(a), None) == -6
    a.stride, a.fold = 1, 9
    assert lib.bsvd_conv3x3(ctypes.byref(a), None) == -7
    a.fold, a.dtype = 0, 7
    assert lib.bsvd_conv3x3(ctypes.byref(a), None) == -2
    assert lib.bsvd_packed_weight_elems(16, 64) == 16 * 9 * 64
    with pytest.raises(ValueError):
        _lib.check(-5, "x")


def test_variant_dry_run_reports_dispatch():
    """bsvd_conv3x3_variant validates like bsvd_conv3x3 and names the kernel instantiation, launching nothing."""
    from bsvd_amd import _lib
    lib = _lib.load()
    buf = ctypes.create_string_buffer(96)

    def variant(**kw):
        a = _lib.BsvdConvArgs()
        a.x = a.y = a.w_packed = 256
        a.frames, a.H, a.W, a.Cin, a.Cout, a.stride = 10, 270, 480, 128, 128, 1
        for k, v in kw.items():
            setattr(a, k, v)
        rc = lib.bsvd_conv3x3_variant(ctypes.byref(a), buf, 96)
        return rc, buf.value.decode()

    assert variant() == (0, "conv3x3_kernel<2,2,2,2,1>[f32]")
    assert variant(fold=16) == (0, "conv3x3_kernel<2,2,2,2,1>[f32]")
    assert variant(fold=12) == (0, "conv3x3_kernel<2,2,2,2,1>[f32][generic]")
    assert variant(dtype=_lib.BSVD_F16X3) == (0, "conv3x3_kernel<4,2,2,2,1>[f16x3]")
    assert variant(dtype=_lib.BSVD_F16X3, frames=1) == (0, "conv3x3_kernel<2,2,2,2,1>[f16x3]")     # small grid: thin tiles
    assert variant(Cout=64, H=540, W=960, Cin=64) == (0, "conv3x3_kernel<2,2,4,1,1>[f32]")
    assert variant(stride=2, Cin=64) == (0, "conv3x3_kernel<2,2,2,2,2>[f32]")
    assert variant(dtype=_lib.BSVD_F16X3, fold=12)[0] == -17 and b"fold 12" in lib.bsvd_last_error()
    # the split stride-2 tile (round 6: 128 px x 32 ch wave tiles) is a plain conv: a temporal shift on it is refused, never read from the wrong frames
    assert variant(dtype=_lib.BSVD_F16X3, stride=2, Cin=64) == (0, "conv3x3_kernel<4,1,1,4,2>[f16x3]")
    assert variant(dtype=_lib.BSVD_F16X3, stride=2, Cin=64, fold=16)[0] == -17 and b"plain convs" in lib.bsvd_last_error()
    # transformed-domain tensors are options of the Winograd form
    assert variant(dtype=_lib.BSVD_F16X3, x_v=6)[0] == -22 and variant(dtype=_lib.BSVD_F16X3, y_v=6)[0] == -22
    assert lib.bsvd_v_groups(240, 6) == 40 and lib.bsvd_v_groups(214, 6) == 40 and lib.bsvd_v_groups(50, 2) == 32 and lib.bsvd_v_groups(240, 5) == -1
    assert lib.bsvd_v_frame_elems(135, 240, 256, 6) == 135 * 5 * 16 * (8 * 32 + 8) * 4 + 135 * 5 * 4 * 256 and lib.bsvd_v_frame_elems(8, 8, 24, 6) == -1
    # a frame of 2 GiB or more cannot be addressed by the split kernel: the error says so (not "fold")
    assert variant(dtype=_lib.BSVD_F16X3, frames=1, H=4320, W=7680, Cin=64, Cout=64)[0] == -17
    assert b"2 GiB" in lib.bsvd_last_error() and b"4320 x 7680" in lib.bsvd_last_error()
    assert variant(frames=1, H=4320, W=7680, Cin=64, Cout=64) == (0, "conv3x3_kernel<2,2,4,1,1>[f32][generic]")
    assert variant(Cin=12)[0] == -5


def test_product_has_no_cpu_fallback():
    """Without a HIP device the product must fail loudly instead of computing on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import bsvd_amd
    m = bsvd_amd.BSVD(precision="fp32", norm="none", pretrain_ckpt=None)
    with pytest.raises(RuntimeError, match="HIP device"):
        m(torch.zeros(1, 2, 4, 8, 8))
    with pytest.raises(RuntimeError, match="HIP device"):
        m.feedin_one_element(torch.zeros(1, 4, 8, 8))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "bsvd_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "conv_ref" not in txt, f


def test_workspace_bytes_is_zero_and_validates():
    """bsvd_workspace_bytes: 0 for valid layers (no scratch in this ABI version), the conv's own error code otherwise."""
    from bsvd_amd import _lib
    lib = _lib.load()
    a = _lib.BsvdConvArgs()
    a.x = a.y = a.w_packed = 16
    a.frames, a.H, a.W, a.Cin, a.Cout, a.stride = 2, 8, 8, 64, 64, 1
    assert lib.bsvd_workspace_bytes(ctypes.byref(a)) == 0
    a.dtype = _lib.BSVD_F16X3
    assert lib.bsvd_workspace_bytes(ctypes.byref(a)) == 0
    a.Cin = 12
    assert lib.bsvd_workspace_bytes(ctypes.byref(a)) == -5
    assert lib.bsvd_workspace_bytes(None) == -1
    assert lib.bsvd_halo_unpack(None, None, 4, 16, 0, 8, 0, None) == -3
    assert lib.bsvd_halo_unpack(16, 16, 4, 16, 12, 8, 0, None) == -3        # c0 + n > C
