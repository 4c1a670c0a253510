"""Shared test helpers (golden loading, reference key lists)."""
import os

import numpy as np

from seeded import seeded_state, state_digest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def bsvd_keys(chns, mid_ch, in_ch, out_ch, interm_ch, blind=False, norm="none"):
    """(key, shape) list of the reference BSVD state_dict in registration order
    (/root/reference/Experimental_root/archs/bsvd_arch.py:116-306, 325-356, 446-450); norm='bn' adds the BatchNorm2d
    tensors where get_norm_function puts them."""
    keys = []

    def conv(k, co, ci):
        keys.append((k + ".weight", (co, ci, 3, 3)))
        keys.append((k + ".bias", (co,)))

    def bn(k, c):
        if norm == "bn":
            keys.extend([(k + ".weight", (c,)), (k + ".bias", (c,)), (k + ".running_mean", (c,)), (k + ".running_var", (c,)),
                         (k + ".num_batches_tracked", ())])

    def mem(k, c):
        conv(k + ".c1.op.conv", c, c)
        bn(k + ".b1", c)
        conv(k + ".c2.op.conv", c, c)
        bn(k + ".b2", c)

    c0, c1, c2 = chns
    for pre, ci, co, bl in (("temp1.", in_ch, mid_ch, blind), ("temp2.", mid_ch, out_ch, False)):
        conv(pre + "inc.convblock.0", interm_ch, 3 if bl else ci)
        bn(pre + "inc.convblock.1", interm_ch)
        conv(pre + "inc.convblock.3", c0, interm_ch)
        bn(pre + "inc.convblock.4", c0)
        conv(pre + "downc0.convblock.0", c1, c0)
        bn(pre + "downc0.convblock.1", c1)
        mem(pre + "downc0.memconv", c1)
        conv(pre + "downc1.convblock.0", c2, c1)
        bn(pre + "downc1.convblock.1", c2)
        mem(pre + "downc1.memconv", c2)
        mem(pre + "upc2.memconv", c2)
        conv(pre + "upc2.convblock.0", 4 * c1, c2)
        mem(pre + "upc1.memconv", c1)
        conv(pre + "upc1.convblock.0", 4 * c0, c1)
        conv(pre + "outc.convblock.0", c0, c0)
        bn(pre + "outc.convblock.1", c0)
        conv(pre + "outc.convblock.3", co, c0)
    return keys


def state_for(g, keys):
    st = seeded_state(keys, int(g["seed"]))
    assert state_digest(st) == str(g["digest"]), "seeded weights differ from the ones the golden was made with"
    return st


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))
