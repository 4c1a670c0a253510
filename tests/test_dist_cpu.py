"""Frame-window sharding over 2 ranks with the gloo backend on CPU: exercises bsvd_amd.dist.HaloExchanger
(the real send/recv protocol) with the oracle-backed executor as the per-rank compute, and checks that the
concatenated shard outputs equal the reference golden of the whole clip."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import load_golden, bsvd_keys, state_for, maxabs

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    for p in (os.path.dirname(HERE), HERE, os.path.join(HERE, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle_exec import OracleExecutor
    from bsvd_amd.dist import HaloExchanger, shard_range
    from bsvd_amd.netspec import make_netspec
    from bsvd_amd.schedule import bsvd_clip
    g = load_golden("g4_bsvd_small_T7")
    st = state_for(g, bsvd_keys([32, 64, 128], 32, 4, 3, 32))
    net = make_netspec([32, 64, 128], 32, 4, 3, "relu6", 32)
    x = torch.from_numpy(g["x"][0])
    a, b = shard_range(x.shape[0], world, rank)
    ex = OracleExecutor(st)
    hx = HaloExchanger(ex, rank, world)
    y = ex.to_nchw(bsvd_clip(ex, net, ex.to_nhwc(x[a:b], 16), hx), 3)
    assert hx.exchanges == 16
    np.save(os.path.join(outdir, "out%d.npy" % rank), y.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_clip_over_gloo(world, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g = load_golden("g4_bsvd_small_T7")
    y = np.concatenate([np.load(tmp_path / ("out%d.npy" % r)) for r in range(world)])
    assert y.shape == g["out"][0].shape
    assert maxabs(y, g["out"][0]) < 1e-4


def test_shard_range_covers_clip():
    from bsvd_amd.dist import shard_range
    for n in (1, 7, 10, 80, 85):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_device_buffer_branch_of_halo_exchange_with_a_stub_backend(monkeypatch):
    """The RCCL branch of HaloExchanger (host_staging False: the backend gets the packed slices / receive buffers
    themselves, finish() only waits) cannot run without two GPUs; its bookkeeping can.  A stub point-to-point fabric
    stands in for torch.distributed: sends and receives are matched by (source, destination) in posting order, the
    payload moves when the RECEIVER's request is waited on, and every wait() is logged.  Two shards of one clip are then
    driven through schedule.bsvd_clip's overlapped form (start -> interior frames -> finish -> boundary frames)."""
    import collections
    import bsvd_amd.dist as D
    from oracle_exec import OracleExecutor
    from bsvd_amd.netspec import make_netspec
    from bsvd_amd.schedule import Halo

    mailbox = collections.defaultdict(collections.deque)        # (src, dst) -> sent tensors
    log = []

    class Op:
        def __init__(self, op, tensor, peer, group=None):
            self.op, self.tensor, self.peer = op, tensor, peer

    class Req:
        def __init__(self, kind, me, op):
            self.kind, self.me, self.op, self.done = kind, me, op, False

        def wait(self):
            log.append((self.me, self.kind, self.op.peer))
            if self.kind == "recv":
                src = mailbox[(self.op.peer, self.me)].popleft()
                assert src.shape == self.op.tensor.shape
                self.op.tensor.copy_(src)
            self.done = True

    current = {"rank": None}

    def batch(ops):
        reqs = []
        for o in ops:
            if o.op == "isend":
                assert o.tensor.is_contiguous()
                mailbox[(current["rank"], o.peer)].append(o.tensor)     # NOT cloned: `keep` must pin it until finish()
                reqs.append(Req("send", current["rank"], o))
            else:
                reqs.append(Req("recv", current["rank"], o))
        return reqs

    monkeypatch.setattr(D.dist, "get_backend", lambda group=None: "nccl")
    monkeypatch.setattr(D.dist, "P2POp", Op)
    monkeypatch.setattr(D.dist, "isend", "isend")
    monkeypatch.setattr(D.dist, "irecv", "irecv")
    monkeypatch.setattr(D.dist, "batch_isend_irecv", batch)

    g = load_golden("g4_bsvd_small_T7")
    st = state_for(g, bsvd_keys([32, 64, 128], 32, 4, 3, 32))
    net = make_netspec([32, 64, 128], 32, 4, 3, "relu6", 32)
    sp = net.temp1["d0c1"]
    ex = OracleExecutor(st)
    torch.manual_seed(3)
    v = torch.rand(6, 8, 12, sp.cin_pad)
    whole = ex.conv(sp, v)
    hx = [D.HaloExchanger(ex, r, 2) for r in (0, 1)]
    assert not hx[0].host_staging
    shards = [v[:3].contiguous(), v[3:].contiguous()]
    pend = []
    for r in (0, 1):
        current["rank"] = r
        pend.append(hx[r].start(sp, shards[r]))
    assert not log, "start() must not wait"
    assert all(p.staged == [] and len(p.keep) == 1 for p in pend)
    outs = []
    for r in (0, 1):
        hp, hn = pend[r].finish()
        assert [e for e in log if e[0] == r] == [(r, "send", 1 - r), (r, "recv", 1 - r)]
        assert pend[r].keep == ()
        assert (hp is None) == (r == 0) and (hn is None) == (r == 1)
        outs.append(ex.conv(sp, shards[r], halo_prev=hp, halo_next=hn))
    assert torch.equal(torch.cat(outs), whole)
    assert hx[0].bytes_sent == 8 * 12 * sp.fold * 4 and hx[0].exchanges == 1
