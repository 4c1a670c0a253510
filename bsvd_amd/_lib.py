"""ctypes binding of libbsvd_hip.so (C ABI in include/bsvd_hip.h).

The library is built in-tree by ``bsvd_amd/csrc/build.sh`` (``__graft_entry__.build()``).  There is
no CPU fallback: if the shared object is missing or a HIP device is absent the product path raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BSVD_HIP_LIB") or os.path.join(_HERE, "libbsvd_hip.so")   # env override: A/B tuning builds

ABI_VERSION = 11
BSVD_F32, BSVD_F16, BSVD_F16X3 = 0, 1, 2
ACT = {"none": 0, "relu": 1, "relu6": 2}
EPI_PLAIN, EPI_PS_ADD, EPI_RESID = 0, 1, 2
BUILD_MEASURE = 1            # bsvd_build_info(): a -DBSVD_MEASURE build (tools/build_measure.sh), never the in-tree product library

EXPORTS = ("bsvd_abi_version", "bsvd_conv_args_size", "bsvd_build_info", "bsvd_last_error", "bsvd_conv3x3", "bsvd_conv3x3_variant", "bsvd_packed_weight_elems", "bsvd_pack_weights",
           "bsvd_packed_head_weight_bytes", "bsvd_pack_head_weights", "bsvd_packed_wino_weight_elems", "bsvd_pack_weights_wino",
           "bsvd_nchw_to_nhwc", "bsvd_nhwc_to_nchw", "bsvd_halo_pack", "bsvd_halo_unpack", "bsvd_workspace_bytes",
           "bsvd_u8_to_planar", "bsvd_planar_to_u8", "bsvd_conv3x3_batch", "bsvd_graph_begin", "bsvd_graph_fork",
           "bsvd_graph_join", "bsvd_graph_end", "bsvd_graph_abort", "bsvd_graph_launch", "bsvd_graph_destroy",
           "bsvd_v_frame_elems", "bsvd_v_groups", "bsvd_to_v")


class BsvdConvArgs(ctypes.Structure):
    """Mirror of ``struct BsvdConvArgs`` (include/bsvd_hip.h) -- keep field order in sync."""
    _fields_ = [
        ("x", ctypes.c_void_p),
        ("x_frame_stride", ctypes.c_int64),
        ("halo_prev", ctypes.c_void_p),
        ("halo_next", ctypes.c_void_p),
        ("halo_prev_pstride", ctypes.c_int32), ("halo_prev_coff", ctypes.c_int32),
        ("halo_next_pstride", ctypes.c_int32), ("halo_next_coff", ctypes.c_int32),
        ("fold", ctypes.c_int32),
        ("w_packed", ctypes.c_void_p),
        ("bias_packed", ctypes.c_void_p),
        ("extra", ctypes.c_void_p),
        ("extra_frame_stride", ctypes.c_int64),
        ("extra_pstride", ctypes.c_int32),
        ("extra_cstride", ctypes.c_int32),
        ("resid_ch", ctypes.c_int32),
        ("y", ctypes.c_void_p),
        ("y_frame_stride", ctypes.c_int64),
        ("frames", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
        ("Cin", ctypes.c_int32), ("Cout", ctypes.c_int32),
        ("stride", ctypes.c_int32),
        ("act", ctypes.c_int32), ("epilogue", ctypes.c_int32), ("dtype", ctypes.c_int32),
        ("x_planar_ch", ctypes.c_int32), ("y_planar_ch", ctypes.c_int32), ("y_clamp", ctypes.c_int32),
        ("y_lo", ctypes.c_float), ("y_hi", ctypes.c_float),
        ("extra_split", ctypes.c_int32),
        ("tile_order", ctypes.c_int32),
        ("head_w_packed", ctypes.c_void_p),
        ("head_bias", ctypes.c_void_p),
        ("w_wino_packed", ctypes.c_void_p),
        ("wino_m", ctypes.c_int32),
        ("fat_min_wgs", ctypes.c_int32),
        ("pre_w_packed", ctypes.c_void_p),
        ("pre_bias", ctypes.c_void_p),
        ("pre_cin", ctypes.c_int32), ("pre_act", ctypes.c_int32),
        ("x_f32", ctypes.c_int32), ("y_f32", ctypes.c_int32),
        ("x_v", ctypes.c_int32), ("y_v", ctypes.c_int32),
    ]


class BsvdLibraryError(RuntimeError):
    pass


_lib = None


def load():
    """Loads the shared object and declares prototypes.  Raises BsvdLibraryError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BsvdLibraryError(
            "libbsvd_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` or "
            "bsvd_amd/csrc/build.sh -- bsvd_amd has no CPU fallback." % LIB_PATH)
    if os.environ.get("BSVD_HIP_LIB"):
        import sys
        print("bsvd_amd: BSVD_HIP_LIB override -- loading %s instead of the in-tree libbsvd_hip.so" % LIB_PATH, file=sys.stderr)
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
    lib.bsvd_abi_version.restype = ctypes.c_int
    lib.bsvd_abi_version.argtypes = []
    lib.bsvd_conv_args_size.restype = ctypes.c_int
    lib.bsvd_conv_args_size.argtypes = []
    lib.bsvd_build_info.restype = ctypes.c_int
    lib.bsvd_build_info.argtypes = []
    lib.bsvd_last_error.restype = ctypes.c_char_p
    lib.bsvd_last_error.argtypes = []
    lib.bsvd_conv3x3.restype = ctypes.c_int
    lib.bsvd_conv3x3.argtypes = [ctypes.POINTER(BsvdConvArgs), vp]
    lib.bsvd_conv3x3_variant.restype = ctypes.c_int
    lib.bsvd_conv3x3_variant.argtypes = [ctypes.POINTER(BsvdConvArgs), ctypes.c_char_p, i32]
    lib.bsvd_packed_weight_elems.restype = i64
    lib.bsvd_packed_weight_elems.argtypes = [i32, i32]
    lib.bsvd_pack_weights.restype = ctypes.c_int
    lib.bsvd_pack_weights.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]
    lib.bsvd_packed_wino_weight_elems.restype = i64
    lib.bsvd_packed_wino_weight_elems.argtypes = [i32, i32, i32]
    lib.bsvd_pack_weights_wino.restype = ctypes.c_int
    lib.bsvd_pack_weights_wino.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]
    lib.bsvd_packed_head_weight_bytes.restype = i64
    lib.bsvd_packed_head_weight_bytes.argtypes = [i32]
    lib.bsvd_pack_head_weights.restype = ctypes.c_int
    lib.bsvd_pack_head_weights.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
    lib.bsvd_nchw_to_nhwc.restype = ctypes.c_int
    lib.bsvd_nchw_to_nhwc.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.bsvd_nhwc_to_nchw.restype = ctypes.c_int
    lib.bsvd_nhwc_to_nchw.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, f32, vp]
    lib.bsvd_u8_to_planar.restype = ctypes.c_int
    lib.bsvd_u8_to_planar.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, f32, vp]
    lib.bsvd_planar_to_u8.restype = ctypes.c_int
    lib.bsvd_planar_to_u8.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.bsvd_halo_pack.restype = ctypes.c_int
    lib.bsvd_halo_pack.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
    lib.bsvd_halo_unpack.restype = ctypes.c_int
    lib.bsvd_halo_unpack.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
    lib.bsvd_conv3x3_batch.restype = ctypes.c_int
    lib.bsvd_conv3x3_batch.argtypes = [ctypes.POINTER(BsvdConvArgs), i32, vp]
    for fn, at in (("bsvd_graph_begin", [vp]), ("bsvd_graph_fork", [vp, vp]), ("bsvd_graph_join", [vp, vp]),
                   ("bsvd_graph_end", [vp, ctypes.POINTER(vp), ctypes.POINTER(i32)]), ("bsvd_graph_abort", [vp]),
                   ("bsvd_graph_launch", [vp, vp]), ("bsvd_graph_destroy", [vp])):
        getattr(lib, fn).restype = ctypes.c_int
        getattr(lib, fn).argtypes = at
    lib.bsvd_v_frame_elems.restype = i64
    lib.bsvd_v_frame_elems.argtypes = [i32, i32, i32, i32]
    lib.bsvd_v_groups.restype = i32
    lib.bsvd_v_groups.argtypes = [i32, i32]
    lib.bsvd_to_v.restype = ctypes.c_int
    lib.bsvd_to_v.argtypes = [vp, i64, i32, vp, i64, i32, i32, i32, i32, i32, vp]
    lib.bsvd_workspace_bytes.restype = i64
    lib.bsvd_workspace_bytes.argtypes = [ctypes.POINTER(BsvdConvArgs)]
    if lib.bsvd_abi_version() != ABI_VERSION:
        raise BsvdLibraryError("libbsvd_hip.so ABI version %d, expected %d" % (lib.bsvd_abi_version(), ABI_VERSION))
    if lib.bsvd_conv_args_size() != ctypes.sizeof(BsvdConvArgs):
        raise BsvdLibraryError("BsvdConvArgs layout mismatch: library %d bytes, binding %d bytes"
                               % (lib.bsvd_conv_args_size(), ctypes.sizeof(BsvdConvArgs)))
    _lib = lib
    return lib


def check(rc, what):
    if rc == 0:
        return
    lib = load()
    if rc < 0:
        raise ValueError("%s: %s (rc=%d)" % (what, lib.bsvd_last_error().decode(), rc))
    raise BsvdLibraryError("%s: HIP error %d" % (what, rc))
