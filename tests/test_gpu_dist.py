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


def _worker(rank, world, port, outdir, precision, kw=None):
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
                      precision=precision, **(kw or {}))
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


def test_two_process_sharding_with_transformed_domain_halos(tmp_path):
    """the same two real processes with F(6,3) layers handing their tensors over in the transformed domain (engine.VT): the halo slices that travel
    between the ranks are block slices of transformed frames (HaloExchanger packs, stages and receives them through the executor)"""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "f16x3", dict(wide_conv="wino6", v_handover=True)), nprocs=world, join=True)
    y = np.concatenate([np.load(tmp_path / ("out%d.npy" % r)) for r in range(world)])
    whole = np.load(tmp_path / "whole.npy")
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


def test_bench_gpus_2_line_on_one_device():
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one rank per GPU), with both ranks on cuda:0
    (BSVD_BENCH_ONE_DEVICE=1: the test knob of a 1-GPU box).  RCCL refuses two ranks on one device, so the halo slices fall back to
    host-staged gloo and the line must SAY so; everything else -- the frame-window sharding, 16 exchanges per forward on both ranks,
    equal bytes, the JSON schema -- is what the first real 8-GPU run will print."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = dict(os.environ, BSVD_BENCH_ONE_DEVICE="1", PYTHONDONTWRITEBYTECODE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--prewarm-s", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["unit"] == "frames/s"
    assert d["config"]["baseline_config"] == "c1 x2" and d["config"]["frames_per_gpu"] == 10 and d["config"]["parallelism"] == "frame-window x2"
    assert d["value"] > 0 and abs(d["value"] - 2 * 10 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    # two ranks on one device cannot open an RCCL communicator: the fallback is taken on ALL ranks and named, with the RCCL error
    assert d["degraded"] is True and d["config"]["halo_transport"].startswith("gloo host-staged (RCCL probe failed")
    per_rank = d["halo"]["per_rank"]
    assert [p["rank"] for p in per_rank] == [0, 1]
    forwards = per_rank[0]["forwards"]
    assert forwards >= 3 and all(p["exchanges"] == 16 * p["forwards"] and p["forwards"] == forwards for p in per_rank)
    assert per_rank[0]["bytes_sent"] == per_rank[1]["bytes_sent"] > 0 and all(p["host_staged"] for p in per_rank)
    assert "sustained" not in d and "power" not in d           # N = 1 only


def test_c4_clip_through_8_ranks_on_one_device_equals_the_unsharded_clip_by_digest():
    """BASELINE config 4 (one 80-frame 540x960 clip) as the driver will launch it at N = 8 -- here with all 8 ranks on cuda:0, halos host-staged
    (a path exercise, `degraded`, not a scaling number) -- against the same clip unsharded at N = 1 (`--scaling strong --total-frames 80`):
    the two lines' per-block output digests must be equal (the denominator of the first real 1 -> 8 curve, and proof that
    eight frame windows + 16 x 7 x 2 halo exchanges reproduce the whole clip bit for bit at full size)."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    common = ["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--prewarm-s", "0", "--no-power-probe", "--no-box-calibration",
              "--scaling", "strong", "--total-frames", "80", "--output-digest"]
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", MASTER_ADDR="127.0.0.1")
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + common, env=env, capture_output=True, text=True,
                        timeout=900, cwd=root)
    l1 = [l for l in r1.stdout.splitlines() if l.startswith("{")]
    assert r1.returncode == 0 and len(l1) == 1, r1.stdout[-2000:] + r1.stderr[-3000:]
    d1 = json.loads(l1[0])
    env8 = dict(env, BSVD_BENCH_ONE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "8"] + common
    r8 = subprocess.run(cmd, env=env8, capture_output=True, text=True, timeout=1500, cwd=root)
    l8 = [l for l in r8.stdout.splitlines() if l.startswith("{")]
    assert r8.returncode == 0 and len(l8) == 1, r8.stdout[-2000:] + r8.stderr[-3000:]
    d8 = json.loads(l8[0])
    assert d1["config"]["baseline_config"] == d8["config"]["baseline_config"] == "c4"
    assert d1["n_gpus"] == 1 and d8["n_gpus"] == 8 and d8["config"]["frames_per_gpu"] == 10 and d1["config"]["frames_per_gpu"] == 80
    assert d8["degraded"] is True and d1["degraded"] is False
    g1, g8 = d1["output_digest"]["sha256_16_per_10_frame_block"], d8["output_digest"]["sha256_16_per_10_frame_block"]
    assert len(g1) == len(g8) == 8 and g1 == g8, (g1, g8)
    per_rank = d8["halo"]["per_rank"]
    assert [p["rank"] for p in per_rank] == list(range(8)) and all(p["exchanges"] == 16 * p["forwards"] for p in per_rank)
    assert per_rank[0]["bytes_sent"] == per_rank[7]["bytes_sent"] and per_rank[1]["bytes_sent"] == 2 * per_rank[0]["bytes_sent"]
    print("C4 on one device: N=1 %.1f frames/s (unsharded 80-frame clip); 8 host-staged ranks on one GPU %.1f frames/s (degraded); digests equal"
          % (d1["value"], d8["value"]))
