"""Frame-window sharding with the REAL HaloExchanger and the HIP executor in two processes sharing cuda:0 (gloo
backend, device tensors): the closest single-GPU stand-in for `bench.py --gpus 2` (which needs RCCL + 2 GPUs).
Checks bitwise equality with the unsharded clip in both arithmetic modes, and that the overlapped schedule
(interior frames first) is the one taken."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import bsvd_keys
from seeded import seeded_state, seeded_clip

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir, precision):
    for p in (os.path.dirname(HERE), HERE, os.path.join(HERE, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import bsvd_amd
    from bsvd_amd.dist import HaloExchanger, shard_range
    st = seeded_state(bsvd_keys([64, 128, 256], 64, 4, 3, 64), 41)
    m = bsvd_amd.BSVD(chns=[64, 128, 256], mid_ch=64, norm="none", act="relu6", interm_ch=64, pretrain_ckpt=None,
                      precision=precision)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    m = m.cuda()
    x = torch.from_numpy(seeded_clip((1, 7, 4, 32, 48), 42, kind="sigma30"))[0].cuda()
    a, b = shard_range(x.shape[0], world, rank)
    hx = HaloExchanger(m._executor(x.device), rank, world)
    y = m.clip_forward(x[a:b], hx)
    torch.cuda.synchronize()
    assert hx.exchanges == 16
    np.save(os.path.join(outdir, "out%d.npy" % rank), y.cpu().numpy())
    if rank == 0:
        np.save(os.path.join(outdir, "whole.npy"), m.clip_forward(x).cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_two_process_sharding_on_one_gpu(tmp_path, precision):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), precision), nprocs=world, join=True)
    y = np.concatenate([np.load(tmp_path / ("out%d.npy" % r)) for r in range(world)])
    whole = np.load(tmp_path / "whole.npy")
    assert y.shape == whole.shape == (7, 3, 32, 48)
    assert np.array_equal(y, whole), "sharded == unsharded bit for bit (max diff %g)" % np.abs(y - whole).max()


def test_rccl_loopback_world_size_1_drives_the_device_buffer_halo_path():
    """the RCCL point-to-point branch of HaloExchanger on real RCCL with a loopback neighbour
    (tests/rccl_loopback_driver.py): req.wait() stream ordering, pinning of the packed send slices, receive-buffer
    lifetime under the caching allocator.  If this RCCL build refuses self send/recv the exact error is shown and the test
    is skipped (then only an N > 1 run can exercise the branch)."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(HERE, "rccl_loopback_driver.py"), str(_free_port())],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert r.returncode == 0 and lines, (r.returncode, r.stderr[-2000:])
    res = json.loads(lines[-1][len("RESULT "):])
    print("\n[rccl loopback] %s" % json.dumps(res))
    if "error" in res:
        pytest.skip("RCCL loopback not possible here: %s" % res["error"])
    assert res["group_backend"] == "nccl"
    for precision, o in res["loopback"].items():
        assert o["equal"], (precision, o)
        assert o["exchanges"] == 3 * 16 and o["bytes_sent"] > 0
