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
