#!/usr/bin/env python3
"""Subprocess driver of tests/test_gpu_dist.py::test_rccl_loopback_* (TEST INFRASTRUCTURE).

One process, one GPU, world_size 1: the process groups are built exactly like ``bench.init_groups`` (gloo control plane,
``new_group(backend="nccl", device_id=...)`` = RCCL data plane) and ``HaloExchanger``'s NON-staged branch
(bsvd_amd/dist.py: device buffers handed to ``batch_isend_irecv``, ``req.wait()`` stream ordering, ``keep`` pinning of the
packed send slices, receive buffers from the caching allocator) runs against a loopback neighbour: rank "1 of 3" whose left
and right peers are both this process.  Inside one batch the two sends meet the two receives in posting order, so the halo
received from the "right" neighbour is this shard's own last-frame slice and the one from the "left" its first-frame slice
-- known payloads.  A whole DenBlock-sized clip then runs through ``clip_forward`` with these exchanges (16 layers x 4
messages on real RCCL) and must equal the same clip run with halos packed directly.

Prints one line ``RESULT {json}``.  If this RCCL build rejects self point-to-point, the exact error is reported instead.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), HERE, os.path.join(HERE, "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch
import torch.distributed as dist


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29577")
    res = {"torch": torch.__version__, "hip": torch.version.hip}
    dist.init_process_group("gloo", rank=0, world_size=1)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    try:
        group = dist.new_group(backend="nccl", device_id=dev)
        res["group_backend"] = dist.get_backend(group)
    except Exception as e:  # noqa: BLE001
        res["error"] = "new_group(backend='nccl'): %s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")
        print("RESULT " + json.dumps(res))
        return
    import bsvd_amd
    from bsvd_amd.dist import HaloExchanger
    from bsvd_amd.schedule import Halo
    from helpers import bsvd_keys
    from seeded import seeded_state, seeded_clip

    class Loopback(HaloExchanger):
        """rank 1 of 3 whose two neighbours are this process"""
        def _peer(self, group_rank):
            return 0

    class Direct:
        """the same halos without any transport: what the loopback exchange must deliver"""
        def __init__(self, ex):
            self.ex = ex

        def __call__(self, spec, v):
            fold = spec.fold
            if fold == 0:
                return None, None
            return (Halo(self.ex.halo_pack(v[0], 0, fold), fold, 0),          # "left" delivers my first frame's [0:fold]
                    Halo(self.ex.halo_pack(v[-1], fold, fold), fold, 0))      # "right" delivers my last frame's [fold:2fold]

    try:
        out = {}
        for precision in ("fp32", "f16x3"):
            st = seeded_state(bsvd_keys([64, 128, 256], 64, 4, 3, 64), 41)
            m = bsvd_amd.BSVD(chns=[64, 128, 256], mid_ch=64, norm="none", act="relu6", interm_ch=64, pretrain_ckpt=None,
                              precision=precision)
            m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
            m = m.cuda()
            x = torch.from_numpy(seeded_clip((1, 5, 4, 64, 96), 42, kind="sigma30"))[0].cuda()
            ex = m._executor(dev)
            hx = Loopback(ex, rank=1, world=3, group=group)
            assert not hx.host_staging, "the RCCL group must take the device-buffer branch"
            want = m.clip_forward(x, Direct(ex))
            torch.cuda.synchronize()
            got = None
            for _ in range(3):                    # repeated: recv buffers are recycled by the caching allocator between rounds
                got = m.clip_forward(x, hx)
                # a burst of allocations right behind the exchange: a receive buffer freed too early would be overwritten
                junk = [torch.full((64, 96, 32), float("nan"), device=dev) for _ in range(8)]
                del junk
            torch.cuda.synchronize()
            out[precision] = {"equal": bool(torch.equal(got, want)), "exchanges": hx.exchanges, "bytes_sent": hx.bytes_sent,
                              "max_abs": float((got - want).abs().max())}
        res["loopback"] = out
    except Exception as e:  # noqa: BLE001
        res["error"] = "%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")
    print("RESULT " + json.dumps(res))
    try:
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass


if __name__ == "__main__":
    main()
