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
