"""uint8 frame I/O on the device: upload frames as uint8 (4x less PCIe traffic than fp32), build the
network input (RGB/255 + constant sigma map, the tensor ``temp_denoise`` hands to the model,
/root/reference/Experimental_root/models/validation_seq_infer.py:15-24) and turn the result into uint8 with the
reference's clamp + round (``tensor2img``, /root/reference/BasicSR/basicsr/utils/img_util.py:66,87-90) -- both on the GPU."""
import ctypes

import torch

from . import _lib
from .engine import require_hip


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def frames_to_input(frames_u8, sigma=None, hwc=True):
    """frames_u8: uint8 device tensor [T,H,W,3] (hwc) or [T,3,H,W] -> fp32 [T,3(+1),H,W] in [0,1]; with ``sigma`` (noise
    std in [0,1] units, e.g. 30/255) a constant noise-map channel is appended."""
    lib = require_hip()
    if frames_u8.dtype != torch.uint8 or not frames_u8.is_cuda or frames_u8.dim() != 4:
        raise ValueError("expected a uint8 device tensor [T,H,W,C] or [T,C,H,W]")
    x = frames_u8.contiguous()
    T, H, W, C = x.shape if hwc else (x.shape[0], x.shape[2], x.shape[3], x.shape[1])
    extra = 0 if sigma is None else 1
    y = torch.empty((T, C + extra, H, W), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.bsvd_u8_to_planar(x.data_ptr(), y.data_ptr(), T, C, H, W, 1 if hwc else 0, extra,
                                         float(sigma or 0.0), _stream()), "bsvd_u8_to_planar")
    return y


def output_to_frames(y, hwc=True, rgb2bgr=False):
    """y: fp32 device tensor [T,C,H,W] -> uint8 [T,H,W,C] (or [T,C,H,W]): clamp [0,1], x255, round half to even."""
    lib = require_hip()
    if y.dtype != torch.float32 or not y.is_cuda or y.dim() != 4:
        raise ValueError("expected a float32 device tensor [T,C,H,W]")
    y = y.contiguous()
    T, C, H, W = y.shape
    out = torch.empty((T, H, W, C) if hwc else (T, C, H, W), dtype=torch.uint8, device=y.device)
    with torch.cuda.device(y.device):
        _lib.check(lib.bsvd_planar_to_u8(y.data_ptr(), out.data_ptr(), T, C, H, W, 1 if hwc else 0, 1 if rgb2bgr else 0,
                                         _stream()), "bsvd_planar_to_u8")
    return out
