"""Checkpoint ingest: accepts the reference's TSN/WNet-schema ``.pth`` (what the authors ship as
bsvd-64.pth) and plain BSVD state_dicts.

Restates the key re-map of ``BSVD.load`` and the per-block ``load`` methods
(/root/reference/Experimental_root/archs/bsvd_arch.py:462-474, 143-145, 252-255, 280-282):

    [module.]base_model.nets_list.{0,1}.X            -> temp{1,2}.X
    downcN.convblock.3.c{1,2}.net.*                   -> downcN.memconv.c{1,2}.op.conv.*
    upcN.convblock.0.c{1,2}.net.*                     -> upcN.memconv.c{1,2}.op.conv.*
    upcN.convblock.1.*                                -> upcN.convblock.0.*
    inc.*, outc.*, downcN.convblock.0.*               unchanged
"""
from collections import OrderedDict

import torch

_STAGE = "base_model.nets_list."


def is_tsn_schema(state):
    return any(_STAGE in k for k in state)


def tsn_key_to_bsvd(key):
    k = key[len("module."):] if key.startswith("module.") else key
    if not k.startswith(_STAGE):
        return None
    stage, rest = k[len(_STAGE):].split(".", 1)
    block, tail = rest.split(".", 1)
    if block.startswith("downc") and tail.startswith("convblock.3."):
        tail = "memconv." + tail[len("convblock.3."):].replace(".net.", ".op.conv.")
    elif block.startswith("upc"):
        if tail.startswith("convblock.0."):
            tail = "memconv." + tail[len("convblock.0."):].replace(".net.", ".op.conv.")
        elif tail.startswith("convblock.1."):
            tail = "convblock.0." + tail[len("convblock.1."):]
    return "temp%d.%s.%s" % (int(stage) + 1, block, tail)


def to_bsvd_state(state):
    """Any supported schema -> OrderedDict in BSVD key names."""
    if not is_tsn_schema(state):
        return OrderedDict((k[len("module."):] if k.startswith("module.") else k, v) for k, v in state.items())
    out = OrderedDict()
    for k, v in state.items():
        nk = tsn_key_to_bsvd(k)
        if nk is not None:
            out[nk] = v
    return out


def fold_batchnorm(state, layer_keys, norm_key_after, eps=1e-5):
    """Eval-mode BatchNorm2d folded into the preceding conv (one-time, at weight-pack time): with
    s = gamma / sqrt(running_var + eps):  w' = w * s[:, None, None, None],  b' = (b - running_mean) * s + beta.
    Covers the reference's default ``norm='bn'`` (get_norm_function, bsvd_arch.py:176-183; BN sits right after the conv,
    before the activation, :122-130, 207-216, 237-241, 294-298).  ``eps``: a float, or {norm key: that BatchNorm2d's eps}.
    Returns a state with conv keys only."""
    eps_of = (lambda k: eps.get(k, 1e-5)) if isinstance(eps, dict) else (lambda k: eps)
    out = OrderedDict()
    for key in layer_keys:
        w, b = state[key + ".weight"].detach().float(), state[key + ".bias"].detach().float()
        nk = norm_key_after(key)
        if nk is not None and nk + ".running_mean" in state:
            mean, var = state[nk + ".running_mean"].detach().float(), state[nk + ".running_var"].detach().float()
            gamma, beta = state[nk + ".weight"].detach().float(), state[nk + ".bias"].detach().float()
            s_ = (gamma.double() / torch.sqrt(var.double() + eps_of(nk)))
            w = (w.double() * s_[:, None, None, None]).float()
            b = ((b.double() - mean.double()) * s_ + beta.double()).float()
        out[key + ".weight"], out[key + ".bias"] = w, b
    return out
