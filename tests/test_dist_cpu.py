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


def _worker80(rank, world, port, outdir):
    for p in (os.path.dirname(HERE), HERE, os.path.join(HERE, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle_exec import OracleExecutor
    from bsvd_amd.dist import HaloExchanger, shard_range
    from bsvd_amd.netspec import make_netspec
    from bsvd_amd.schedule import bsvd_clip
    g = load_golden("g4_bsvd_small_T7")
    st = state_for(g, bsvd_keys([32, 64, 128], 32, 4, 3, 32))
    net = make_netspec([32, 64, 128], 32, 4, 3, "relu6", 32)
    x = torch.from_numpy(np.load(os.path.join(outdir, "x80.npy")))
    a, b = shard_range(x.shape[0], world, rank)
    assert b - a == 10                                   # BASELINE config 4: 80 frames, 10 per rank
    ex = OracleExecutor(st)
    hx = HaloExchanger(ex, rank, world)
    y = ex.to_nchw(bsvd_clip(ex, net, ex.to_nhwc(x[a:b], 16), hx), 3)
    assert hx.exchanges == 16                            # one per temporal-fusion layer, interior ranks on both sides
    np.save(os.path.join(outdir, "out%d.npy" % rank), y.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_c4_80_frames_over_8_real_gloo_ranks(tmp_path):
    """BASELINE config 4's split -- one 80-frame clip, 8 ranks x 10 frames, a 1-frame halo per temporal-fusion layer and boundary -- as
    EIGHT REAL PROCESSES over gloo (world 2 / 3 above, the stub fabric below at 3 / 8): the concatenated shard outputs
    equal the unsharded clip of the same executor to fp32 rounding (16 x 2 messages per interior rank; ranks 0 and 7 have one neighbour).
    (The CPU executor's torch.conv2d picks batch- and thread-dependent kernels, so 10-frame and 80-frame calls differ in the last bit;
    BIT equality of the sharded schedule is the GPU suite's test_gpu_fullsize.py::test_80_frames_in_8_shards on the HIP kernels.)"""
    from oracle_exec import OracleExecutor
    from bsvd_amd.netspec import make_netspec
    from bsvd_amd.schedule import bsvd_clip
    g = load_golden("g4_bsvd_small_T7")
    st = state_for(g, bsvd_keys([32, 64, 128], 32, 4, 3, 32))
    net = make_netspec([32, 64, 128], 32, 4, 3, "relu6", 32)
    rs = np.random.RandomState(80)
    x = rs.rand(80, 4, 8, 12).astype(np.float32)
    np.save(tmp_path / "x80.npy", x)
    port = _free_port()
    mp.spawn(_worker80, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    y = np.concatenate([np.load(tmp_path / ("out%d.npy" % r)) for r in range(8)])
    ex = OracleExecutor(st)
    want = ex.to_nchw(bsvd_clip(ex, net, ex.to_nhwc(torch.from_numpy(x), 16), None), 3).numpy()
    assert y.shape == want.shape == (80, 3, 8, 12)
    assert maxabs(y, want) < 2e-5, maxabs(y, want)
    # a wrong or missing halo is not a rounding matter: without any exchange the shard boundaries are off by orders of magnitude more
    lone = np.concatenate([ex.to_nchw(bsvd_clip(ex, net, ex.to_nhwc(torch.from_numpy(x[a:a + 10]), 16), None), 3).numpy() for a in range(0, 80, 10)])
    assert maxabs(lone, want) > 1e-2


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


@pytest.mark.parametrize("world", [3, 8])
def test_edge_and_interior_ranks_pair_up_in_posting_order_on_a_stub_fabric(monkeypatch, world):
    """What only N > 2 has: the outermost ranks post 2 point-to-point ops per temporal-fusion layer, interior ranks 4 (right
    neighbour first, then left), all inside one batch_isend_irecv.  A stub fabric with NCCL's matching rule -- between a pair of
    ranks, the k-th send of one meets the k-th receive of the other, in posting order -- carries the device-buffer branch of
    HaloExchanger (host_staging False) for world 3 and 8, two consecutive layers (a second exchange must not pick up the first
    one's slices): every rank's output equals its window of the unsharded layer, op counts and bytes are what bench.py's per-rank
    `halo` block reports (the reference cannot shard at all: validation_seq_infer.py:24)."""
    import collections
    import bsvd_amd.dist as D
    from oracle_exec import OracleExecutor
    from bsvd_amd.netspec import make_netspec

    mailbox = collections.defaultdict(collections.deque)        # (src, dst) -> tensors in send order
    posted = collections.defaultdict(list)                      # rank -> [(kind, peer)] in posting order
    current = {"rank": None}

    class Op:
        def __init__(self, op, tensor, peer, group=None):
            self.op, self.tensor, self.peer = op, tensor, peer

    class Req:
        def __init__(self, kind, me, op):
            self.kind, self.me, self.op = kind, me, op

        def wait(self):
            if self.kind == "recv":
                src = mailbox[(self.op.peer, self.me)].popleft()          # k-th receive from a peer <- that peer's k-th send to me
                assert src.shape == self.op.tensor.shape
                self.op.tensor.copy_(src)

    def batch(ops):
        reqs = []
        for o in ops:
            posted[current["rank"]].append((o.op, o.peer))
            if o.op == "isend":
                mailbox[(current["rank"], o.peer)].append(o.tensor)
            reqs.append(Req("send" if o.op == "isend" else "recv", current["rank"], o))
        return reqs

    monkeypatch.setattr(D.dist, "get_backend", lambda group=None: "nccl")
    monkeypatch.setattr(D.dist, "P2POp", Op)
    monkeypatch.setattr(D.dist, "isend", "isend")
    monkeypatch.setattr(D.dist, "irecv", "irecv")
    monkeypatch.setattr(D.dist, "batch_isend_irecv", batch)

    g = load_golden("g4_bsvd_small_T7")
    st = state_for(g, bsvd_keys([32, 64, 128], 32, 4, 3, 32))
    net = make_netspec([32, 64, 128], 32, 4, 3, "relu6", 32)
    sp1, sp2 = net.temp1["d0c1"], net.temp1["d0c2"]
    ex = OracleExecutor(st)
    torch.manual_seed(5)
    T = 2 * world + 1                                            # ragged: one rank owns 3 frames
    v = torch.rand(T, 8, 12, sp1.cin_pad)
    whole1 = ex.conv(sp1, v)
    whole2 = ex.conv(sp2, whole1)
    spans = [D.shard_range(T, world, r) for r in range(world)]
    hx = [D.HaloExchanger(ex, r, world) for r in range(world)]
    cur = [v[a:b].contiguous() for a, b in spans]
    for sp, whole in ((sp1, whole1), (sp2, whole2)):
        pend = []
        for r in range(world):                                   # every rank posts before anybody waits (overlapped schedule)
            current["rank"] = r
            pend.append(hx[r].start(sp, cur[r]))
        nxt = []
        for r in reversed(range(world)):                         # waits in any order
            hp, hn = pend[r].finish()
            assert (hp is None) == (r == 0) and (hn is None) == (r == world - 1)
            nxt.append((r, ex.conv(sp, cur[r], halo_prev=hp, halo_next=hn)))
        cur = [y for _, y in sorted(nxt)]
        assert torch.equal(torch.cat(cur), whole)
        assert all(len(q) == 0 for q in mailbox.values()), "every sent slice was received exactly once"
    for r in range(world):
        ops = posted[r]
        edge = r in (0, world - 1)
        assert len(ops) == (2 if edge else 4) * 2                # two layers
        per_layer = ops[:len(ops) // 2]
        want = ([("isend", r + 1), ("irecv", r + 1)] if r + 1 < world else []) + ([("isend", r - 1), ("irecv", r - 1)] if r > 0 else [])
        assert per_layer == want and ops[len(ops) // 2:] == want
        assert hx[r].exchanges == 2
        assert hx[r].bytes_sent == (1 if edge else 2) * 8 * 12 * 4 * (sp1.fold + sp2.fold)
