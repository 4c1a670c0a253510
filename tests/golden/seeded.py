"""Seeded synthetic weights / inputs shared by the golden generator and the tests.

Nothing here comes from the reference: weights are regenerated on demand from a
numpy RandomState so that fixtures never have to store them.  The distribution
mirrors the reference initialisation (Kaiming-normal conv weights,
/root/reference/Experimental_root/archs/bsvd_arch.py:476-483; torch's default
uniform bias) so that activations keep an O(1) scale through the 32 convs.
"""
import hashlib
from collections import OrderedDict

import numpy as np


def seeded_state(named_shapes, seed):
    """named_shapes: iterable of (key, shape) in state_dict order -> OrderedDict of float32 arrays."""
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for key, shape in named_shapes:
        shape = tuple(int(s) for s in shape)
        # BatchNorm2d tensors (norm='bn' networks): affine, running statistics and the step counter of a "trained" layer
        if key.endswith(".running_mean"):
            out[key] = (rs.standard_normal(shape) * 0.2).astype(np.float32)
        elif key.endswith(".running_var"):
            out[key] = rs.uniform(0.5, 1.5, size=shape).astype(np.float32)
        elif key.endswith(".num_batches_tracked"):
            out[key] = np.array(7, dtype=np.int64)
        elif key.endswith(".weight") and len(shape) == 1:
            out[key] = rs.uniform(0.5, 1.5, size=shape).astype(np.float32)
        elif key.endswith(".bias") and len(next(reversed(out.values())).shape) == 1:
            out[key] = rs.uniform(-0.2, 0.2, size=shape).astype(np.float32)
        elif key.endswith(".weight"):
            fan_in = shape[1] * shape[2] * shape[3]
            out[key] = (rs.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        elif key.endswith(".bias"):
            # fan_in of the matching weight = previous entry
            prev = next(reversed(out.values()))
            fan_in = prev.shape[1] * prev.shape[2] * prev.shape[3]
            b = 1.0 / np.sqrt(fan_in)
            out[key] = rs.uniform(-b, b, size=shape).astype(np.float32)
        else:
            raise KeyError(key)
    return out


def state_digest(state):
    h = hashlib.sha256()
    for k, v in state.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v, dtype=np.float32).tobytes())
    return h.hexdigest()


def seeded_clip(shape, seed, kind="randn"):
    """Synthetic input clip [N,F,C,H,W] float32.

    kind='randn'  : profile.py-style gaussian input (S1 of SURVEY §8d)
    kind='sigma30': smooth-ish clean clip + AWGN sigma=30/255, 4th channel = constant noise map (S2)
    """
    rs = np.random.RandomState(seed)
    if kind == "randn":
        return rs.standard_normal(shape).astype(np.float32)
    n, f, c, h, w = shape
    assert c in (3, 4)
    gt = rs.uniform(0.0, 1.0, size=(n, f, 3, h, w)).astype(np.float32)
    lq = gt + rs.standard_normal(gt.shape).astype(np.float32) * np.float32(30.0 / 255.0)
    if c == 3:
        return lq.astype(np.float32)
    nm = np.full((n, f, 1, h, w), 30.0 / 255.0, dtype=np.float32)
    return np.concatenate([lq, nm], axis=2).astype(np.float32)
