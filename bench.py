#!/usr/bin/env python3
"""bench.py -- throughput of the BSVD hot path on MI355X (driver contract in the task statement).

A "step" = one ``BSVD.forward`` pass over one synthetic sigma=30 clip of ``--frames`` (10) frames of
540x960 per GPU (BASELINE.json configs: bsvd_c64, synthetic [1,10,4,540,960]; profile.py protocol of the
reference: input already resident on the device, /root/reference/profile.py:70-83).  With N GPUs the job
is ONE clip of 10*N frames, frame-window sharded, every temporal-fusion layer swapping a 1-frame halo with
its neighbours over RCCL (bsvd_amd/dist.py) -> weak scaling; N=8 is BASELINE config 4 (80 frames).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Other BASELINE configurations (same JSON schema; the default above is what the driver runs):
    --workload c2   bsvd_c64 sigma=30, DAVIS-2017-test-dev-480p geometry: one 85-frame 480x856 clip
    --workload c3   blind bsvd_c64 (3-channel input, interm_ch 30, ReLU; WNet semantics), Set8 geometry: 85 x 540x960
    --workload c5   bsvd_c64 1080p (1920x1080) streaming: 64 frames through feedin_one_element (--mode perframe)
    --mode clip | stream (streaming_forward: rings + HIP graphs, chunked) | perframe (feedin_one_element, one graph per frame)
    --scaling strong --total-frames 80   one fixed clip split over the ranks (N=1 and N=8 run the same C4 clip)

Prints ONE JSON line on rank 0.  ``value`` = total frames / max-over-ranks wall time of the K timed steps.
``roofline``: per-launch HIP-event timing (on the stream the kernels run on) of the dominant kernel
against the fp32-MFMA peak.  ``cpu_baseline``: the oracle's streaming restatement (same oneDNN conv calls
as the reference's CPU forward) timed on this box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

REFERENCE_PUBLISHED_FPS = 10 / 0.353594
PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense fp16 MFMA peak (same guide); f16x3 issues 3 MFMA FLOPs per algorithmic FLOP
DEFAULT_PRECISION = "f16x3"
H, W = 540, 960
SIGMA = 30.0 / 255.0

# BASELINE.json configs: geometry, frames per GPU per step, default schedule, network variant
WORKLOADS = {
    "c1": dict(h=540, w=960, frames=10, mode="clip", blind=False,
               name="C1/C4 bsvd_c64 sigma=30, synthetic clip [1,%d,4,540,960]"),
    "c2": dict(h=480, w=856, frames=85, mode="clip", blind=False,
               name="C2 bsvd_c64 sigma=30, DAVIS-2017-test-dev-480p geometry (854x480 padded to 856x480), one %d-frame clip"),
    "c3": dict(h=540, w=960, frames=85, mode="clip", blind=True,
               name="C3 blind bsvd_c64 (3-channel input, interm_ch 30, ReLU, WNet semantics), Set8 geometry, one %d-frame 540x960 clip"),
    "c5": dict(h=1080, w=1920, frames=64, mode="perframe", blind=False,
               name="C5 bsvd_c64 sigma=30 1080p (1920x1080) streaming, %d frames"),
}


def synth_clip(frames, seed, device, h=None, w=None):
    """S2 of SURVEY §8d: clean uniform clip + AWGN sigma=30/255, 4th channel = constant noise map."""
    h, w = h or H, w or W
    g = torch.Generator(device="cpu").manual_seed(seed)
    gt = torch.rand((1, frames, 3, h, w), generator=g)
    lq = gt + torch.randn(gt.shape, generator=g) * SIGMA
    nm = torch.full((1, frames, 1, h, w), SIGMA)
    return lq.to(device), nm.to(device)


def synth_window(start, end, block, device, h, w, blind):
    """frames [start, end) of the clip whose block b (frames [b * block, (b + 1) * block)) is synth_clip(block, 100 + b): [F,C,h,w]"""
    parts = []
    for b in range(start // block, (end - 1) // block + 1):
        lq, nm = synth_clip(block, 100 + b, device, h, w)
        v = (lq if blind else torch.cat([lq, nm], dim=2))[0]
        lo, hi = max(start, b * block) - b * block, min(end, (b + 1) * block) - b * block
        parts.append(v[lo:hi])
    return torch.cat(parts, dim=0).contiguous() if len(parts) > 1 else parts[0].contiguous()


def output_digests(y, first_frame, block=10):
    """sha256 (16 hex digits) of every whole `block`-frame block of the job's output that lies inside this rank's window [first_frame, ..):
    {block index: digest}.  Outside every timed region."""
    import hashlib
    out = {}
    n = y.shape[0]
    for b in range((first_frame + block - 1) // block, (first_frame + n) // block):
        lo = b * block - first_frame
        out[b] = hashlib.sha256(y[lo:lo + block].contiguous().cpu().numpy().tobytes()).hexdigest()[:16]
    return out


def build_model(device, precision="fp32", blind=False, wide_conv="auto", fuse_pairs="auto", f32_handover="auto", v_handover="auto"):
    import bsvd_amd
    torch.manual_seed(1234)      # random-init weights of the bsvd_c64 architecture (no checkpoint in the tree)
    if blind:                    # options/test/0407...blind_c64.yml:97-121: interm_ch default 30, act default relu
        m = bsvd_amd.BSVD(chns=[64, 128, 256], mid_ch=64, shift_input=False, in_ch=4, out_ch=3, norm="none",
                          act="relu", interm_ch=30, blind=True, pretrain_ckpt=None, precision=precision, wide_conv=wide_conv,
                          fuse_pairs=fuse_pairs, f32_handover=f32_handover, v_handover=v_handover)
    else:
        m = bsvd_amd.BSVD(chns=[64, 128, 256], mid_ch=64, shift_input=False, in_ch=4, out_ch=3, norm="none",
                          act="relu6", interm_ch=64, blind=False, pretrain_ckpt=None, precision=precision, wide_conv=wide_conv,
                          fuse_pairs=fuse_pairs, f32_handover=f32_handover, v_handover=v_handover)
    return m.to(device).eval()


class _Both:
    """FLOP accounting of a fused pair of layers"""

    def __init__(self, *sps):
        self.sps = sps

    def macs(self, h, w):
        return sum(sp.macs(h, w) for sp in self.sps)


def algorithmic_bytes(sp, T, h, w, x_planar_ch=0, y_planar_ch=0, has_extra=False):
    """HBM bytes ONE launch of layer ``sp`` has to move at the very least: every input and output element
    once at the 4 bytes both arithmetic modes store per value (fp32 / an fp16 pair), the packed weights once, the PixelShuffle skip
    tensor / the residual base once.  The temporal-shift gather reads fold channels of the neighbour frames INSTEAD of the frame's
    own, so it adds nothing.  A fused pair (_Both) never moves its intermediate tensor."""
    if isinstance(sp, _Both):
        first, last = sp.sps[0], sp.sps[-1]
        wbytes = sum(q.cin_pad * 9 * q.cout_pad * 4 for q in sp.sps)
        return T * h * w * 4 * ((x_planar_ch or first.cin_pad) + last.cout_pad) + wbytes
    ho, wo = (h - 1) // sp.stride + 1, (w - 1) // sp.stride + 1
    n_in = T * h * w * (x_planar_ch or sp.cin_pad)
    n_out = T * ho * wo * (y_planar_ch or sp.cout_pad)           # PixelShuffle: 4 x the pixels at a quarter of the channels
    n_extra = 0
    if has_extra:
        n_extra = n_out if sp.epilogue == 1 else T * ho * wo * min(3, sp.cout)
    return 4 * (n_in + n_out + n_extra) + sp.cin_pad * 9 * sp.cout_pad * 4


def mfma_factor(kernel_name):
    """MFMA flop issued per algorithmic flop by a kernel variant (name as bsvd_conv3x3_variant reports it)"""
    import re
    passes = 3.0 if "f16x3" in kernel_name else 1.0
    if "[fused pair]" in kernel_name:       # the first conv runs on 12 MFMA row tiles (384 pixel slots) per 256 output pixels: 1.5x its share (half of an equal pair)
        passes *= 1.25
    m = re.search(r"F\((\d),3\)", kernel_name)                   # Winograd F(m,3) along x: 3 (m + 2) tap-GEMMs per m outputs instead of 9 per output
    return passes * ((m_ := int(m.group(1))) + 2) * 3.0 / (9.0 * m_) if m else passes


class LaunchTimer:
    """Wraps HipExecutor.conv with a HIP event pair per launch (same stream as the kernel).  Events come from a pool
    that is filled during the warmup steps and re-recorded in the timed region, and the kernel-variant names are cached
    per (layer, shape), so the timed region pays two hipEventRecord per launch and nothing else."""

    def __init__(self, ex):
        self.ex = ex
        self.records = []
        self.pool, self.used = [], 0
        self.names = {}
        self._orig = ex.conv
        ex.conv = self._conv
        self._orig_fused = ex.conv_head_fused
        ex.conv_head_fused = self._fused
        self._orig_pair = ex.conv_pair_fused
        ex.conv_pair_fused = self._pair

    def _event(self):
        if self.used == len(self.pool):
            self.pool.append(torch.cuda.Event(enable_timing=True))
        self.used += 1
        return self.pool[self.used - 1]

    def _conv(self, sp, x, *a, **k):
        key = (sp.key, tuple(x.shape), bool(k.get("x_planar")), bool(k.get("y_planar")))
        name = self.names.get(key)
        self.ex.record_variants = name is None   # first sight: ask the library (bsvd_conv3x3_variant) what it dispatches
        e0, e1 = self._event(), self._event()
        e0.record()
        y = self._orig(sp, x, *a, **k)
        e1.record()
        if name is None:
            name = self.names[key] = self.ex.last_variant
        if k.get("x_planar"):
            T, _, Hh, Ww = x.shape
        else:
            T, Hh, Ww, _ = x.shape
        # temporal-shift chunks the kernel leaves out of K (zero groups of a clip's first / last frame when no halo is supplied)
        zs = 0
        if getattr(sp, "tsm", False) and sp.fold % 16 == 0:
            zs = (0 if (len(a) > 0 and a[0] is not None) or k.get("halo_prev") is not None else 1) + \
                 (0 if (len(a) > 1 and a[1] is not None) or k.get("halo_next") is not None else 1)
        yp = k.get("y_planar")
        extra = k.get("extra") if "extra" in k else (a[2] if len(a) > 2 else None)
        nbytes = algorithmic_bytes(sp, T, Hh, Ww, x.shape[1] if k.get("x_planar") else 0, yp[0] if yp else 0, extra is not None)
        self.records.append((sp, T, Hh, Ww, e0, e1, name, zs, nbytes))
        return y

    def _fused(self, sp0, sp3, x, *a, **k):
        """InputCvBlock as one launch (engine.head_fusable): its algorithmic FLOP are the two convs'"""
        key = (sp3.key, tuple(x.shape), "fused")
        name = self.names.get(key)
        self.ex.record_variants = name is None
        e0, e1 = self._event(), self._event()
        e0.record()
        y = self._orig_fused(sp0, sp3, x, *a, **k)
        e1.record()
        if name is None:
            name = self.names[key] = self.ex.last_variant
        T, _, Hh, Ww = x.shape
        both = _Both(sp0, sp3)
        self.records.append((both, T, Hh, Ww, e0, e1, name, 0, algorithmic_bytes(both, T, Hh, Ww, x.shape[1])))
        return y

    def _pair(self, spa, spb, x, *a, **k):
        """a fused pair of plain convs as one launch: its algorithmic FLOP and bytes are the two convs' without the tensor between them"""
        key = (spb.key, tuple(x.shape), "pair", bool(k.get("y_planar")))
        name = self.names.get(key)
        self.ex.record_variants = name is None
        e0, e1 = self._event(), self._event()
        e0.record()
        y = self._orig_pair(spa, spb, x, *a, **k)
        e1.record()
        if name is None:
            name = self.names[key] = self.ex.last_variant
        T, Hh, Ww, _ = x.shape
        both = _Both(spa, spb)
        yp = k.get("y_planar")
        extra = k.get("extra") if "extra" in k else (a[0] if len(a) > 0 else None)
        nb = T * Hh * Ww * 4 * (spa.cin_pad + (yp[0] if yp else spb.cout_pad) + (min(3, spb.cout) if extra is not None else 0)) + \
            (spa.cin_pad * 9 * spa.cout_pad + spb.cin_pad * 9 * spb.cout_pad) * 4
        self.records.append((both, T, Hh, Ww, e0, e1, name, 0, nb))
        return y

    def reserve(self, steps):
        """grow the pool to `steps` x (events used since the last reset), creating the HIP events now (record() creates)"""
        need = steps * max(self.used, 1)
        while len(self.pool) < need:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.pool.append(e)

    def reset(self):
        """forget the recorded launches (warmup), keep the event pool and the name cache"""
        self.records, self.used = [], 0

    def detach(self):
        self.ex.conv = self._orig
        self.ex.conv_head_fused = self._orig_fused
        self.ex.conv_pair_fused = self._orig_pair
        self.ex.record_variants = False

    def summary(self):
        """per kernel variant: time, ALGORITHMIC flop (2 x MACs of the convolution) and the MFMA flop the kernel actually issues:
        x passes (1 exact fp32, 3 split), x the tap-GEMMs of its arithmetic form over the direct form's 9 (Winograd F(m,3) along x:
        3 (m + 2) / m), minus the all-zero temporal-shift chunks it leaves out of K (fold / Cin of one frame's K per missing neighbour)"""
        agg = {}
        for sp, T, Hh, Ww, e0, e1, name, zs, nbytes in self.records:
            ms = e0.elapsed_time(e1)
            d = agg.setdefault(name, {"ms": 0.0, "flop": 0.0, "flop_issued": 0.0, "launches": 0, "bytes": 0.0})
            d["bytes"] += nbytes
            flop = 2.0 * sp.macs(Hh, Ww) * T
            d["ms"] += ms
            d["flop"] += flop
            d["flop_issued"] += flop * mfma_factor(name) * (1.0 - zs * (sp.fold / sp.cin) / T if zs else 1.0)
            d["launches"] += 1
        return agg


def usable_cores():
    """Host cores this process may really use: min(affinity mask, cgroup CPU quota).  The GPU box exposes 256
    logical CPUs but a 16-CPU cgroup quota; oversubscribing it makes oneDNN ~10x slower (tools/cpu_probe.py)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(model, h, w, blind, frames=10):
    """Reference CPU path stand-in: oracle.stream_forward (per-frame pipeline, F.conv2d fp32 + torch.cat,
    the same oneDNN calls as the reference's CPU forward; timed beside the real reference in the build container:
    profiles/cpu_ref_vs_port.json) on the host cores this process may use, on one [1,F,C,h,w] sigma=30 clip:
    one warm-up forward + best of 2."""
    from oracle import bsvd_oracle as O
    P = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    cores = usable_cores()
    torch.set_num_threads(cores)
    lq, nm = synth_clip(frames, 8, "cpu", h, w)
    if blind:
        nm = None
    times = []
    with torch.no_grad():
        for _ in range(3):
            t0 = time.perf_counter()
            O.stream_forward(lq, P, noise_map=nm)
            times.append(time.perf_counter() - t0)
    dt = min(times[1:])
    cpu = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": frames / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "runs_s": times,
            "sample": "oracle.stream_forward (per-frame pipeline, torch conv2d fp32) on one [1,%d,%d,%d,%d] sigma=30 clip, "
                      "1 warm-up + best of 2; host CPU: %s" % (frames, 3 if blind else 4, h, w, cpu)}


def power_probe(step, seconds, allow_empty=False):
    """rocm-smi package power / shader clock while `step` loops (untimed, after the timed region, N = 1 only): the split mode runs at
    the package power cap with the clock throttled, and the line should say so itself.  None if rocm-smi is
    absent or prints something else."""
    import re
    import subprocess
    import threading
    import torch
    rows, stop = [], threading.Event()

    # (a profiler wrapped around bench.py must not follow into the rocm-smi child)
    env = {k: v for k, v in os.environ.items()
           if not (k.startswith(("ROCP", "ROCPROF", "ROCTX")) or k in ("LD_PRELOAD", "HSA_TOOLS_LIB", "HSA_TOOLS_REPORT_LOAD_FAILURE"))}

    def smi(*flags):
        return subprocess.run(["rocm-smi", *flags], capture_output=True, text=True, timeout=10, env=env).stdout

    def sample():
        while not stop.is_set():
            try:
                out = smi("--showpower", "--showclocks")
                pw = re.search(r"GPU\[0\].*?Package Power \(W\):\s*([0-9.]+)", out)
                ck = re.search(r"GPU\[0\].*?sclk clock level:.*?\((\d+)Mhz\)", out)
                if pw and ck:
                    rows.append((float(pw.group(1)), float(ck.group(1))))
            except Exception:                                     # noqa: BLE001 - a measurement aid must never fail the bench
                return
            stop.wait(0.3)

    try:
        cap = re.search(r"Max Graphics Package Power \(W\):\s*([0-9.]+)", smi("--showmaxpower"))
    except Exception:                                             # noqa: BLE001
        return None
    th = threading.Thread(target=sample, daemon=True)
    th.start()
    t0 = time.perf_counter()
    nsteps = 0
    gpu = torch.cuda.is_available()
    e0, e1 = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if gpu else (None, None)
    with torch.no_grad():
        if gpu:
            e0.record()
        while time.perf_counter() - t0 < seconds:
            for _ in range(4):
                step()
            nsteps += 4
            if gpu:
                torch.cuda.synchronize()
        if gpu:
            e1.record()
            torch.cuda.synchronize()
    loop_s = time.perf_counter() - t0
    event_ms = e0.elapsed_time(e1) if gpu else None      # HIP events around the whole loop (bench.box_calibration times its layers with them)
    stop.set()
    th.join(timeout=15)
    rows = rows[1:] if len(rows) > 2 else rows                    # the first sample may predate the loop
    if not rows:
        return {"package_w": None, "cap_w": float(cap.group(1)) if cap else None, "sclk_mhz": None, "samples": 0, "loop_steps": nsteps,
                "loop_seconds": loop_s, "event_ms": event_ms, "source": "no rocm-smi sample landed inside the loop"} if allow_empty else None
    return {"event_ms": event_ms, "package_w": sum(r[0] for r in rows) / len(rows), "cap_w": float(cap.group(1)) if cap else None,
            "sclk_mhz": sum(r[1] for r in rows) / len(rows), "samples": len(rows), "loop_steps": nsteps, "loop_seconds": loop_s,
            "source": "rocm-smi --showpower --showclocks every ~0.5 s while the same step loops for %.1f s after the timed region" % seconds}


# ---- box calibration: two FIXED layers on seeded operands, timed outside the timed region, so that a line says how fast
# ITS box is next to the box the reference figures below come from.  The driver's boxes differ by +-5 % on the same commit (r05: 366.0 on
# the driver's box, 384.4 on the builder's); without this, "did round N beat round N-1" had to be inferred from untouched kernels.
#   direct64 : out0 of DenBlock 1 (64 -> 64 plain conv at 540 x 960, 10 frames), the direct-form MFMA tile <4,1,2,2,1>.  Its device code is
#              byte-identical since round 5 (ISA digest of conv3x3_mfma.hip unchanged by round 6) -- the NORMALISATION BASIS: the kernel
#              a round works on must not normalise its own gain away
#   wino256  : d1c2 of DenBlock 1 (256 -> 256 temporal-fusion conv at 135 x 240, 10 frames) in the shipped form -- informational: what the
#              round's dominant kernel does on this box
# Reference figures: a NOMINAL box = the r05 record box (profiles/r06_box_calibration.json lists the boxes seen so far under this protocol).
BOX_CAL_REF = {"direct64_ms": 1.000, "wino256_ms": 0.631,
               "source": "nominal reference box = the r05 record box (direct64 0.990 ms in its clip, 256->256 F(2,3) 0.631 ms, C1 384.4 frames/s); the r05 "
                         "driver box: direct64 1.038, C1 366.0; the r06 session-1 box: 1.047, C1 366.8 (profiles/r06_box_calibration.json)"}


def kernel_sources_sha16():
    """sha256 (16 hex digits) over the kernel sources + the ABI header of this tree, `//` comments and blank space stripped (a reworded comment is not a new
    kernel) -- what profiles/traffic.json records for the tree its PMC passes ran on"""
    import glob
    import hashlib
    import re
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "bsvd_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "bsvd_amd", "csrc", "*.h")) + [os.path.join(ROOT, "include", "bsvd_hip.h")]):
        h.update(os.path.basename(f).encode())
        for line in open(f, "r", errors="replace"):
            line = re.sub(r"\s+", " ", re.sub(r"//.*$", "", line)).strip()
            if line:
                h.update(line.encode())
    return h.hexdigest()[:16]


def normalise_value(value, cal_ms, ref_ms):
    """frames/s this run would show on the reference box: a box whose fixed calibration layer takes cal_ms where the reference box takes
    ref_ms is cal_ms / ref_ms slower, and throughput scales with the inverse.  None when either figure is missing."""
    if not cal_ms or not ref_ms or cal_ms <= 0 or ref_ms <= 0:
        return None
    return value * (cal_ms / ref_ms)


def box_calibration(model, x, seconds=0.7):
    """ms per launch of the two fixed layers (HIP events around a ~`seconds` loop each, after a short ramp), with the clock and package power
    rocm-smi shows meanwhile.  `x`: the step's own [F,C,H,W] clip -- the 64-channel operand is DenBlock 1's real x0 for its first 10 frames
    (seeded clip + seeded weights: the same bits on every box); the 256-channel operand is a seeded uniform [0, 6) tensor."""
    from bsvd_amd import schedule
    dev = x.device
    ex = model._executor(dev)
    S = model.net.temp1
    out = {}
    with torch.no_grad():
        # FIXED operands whatever the workload: the first 10 frames of the C1 clip at 540 x 960 (the step's own clip when it is that one)
        if tuple(x.shape[-2:]) != (H, W) or x.shape[0] < 10 or x.shape[1] != 4:
            x = synth_window(0, 10, 10, dev, H, W, False)
        x0 = schedule._inc(ex, S, x[:10].contiguous(), True)
        g = torch.Generator(device="cpu").manual_seed(77)
        sp_w = S["d1c2"]
        h4, w4 = (x.shape[-2] + 3) // 4, (x.shape[-1] + 3) // 4
        xw = (torch.rand((10, h4, w4, sp_w.cin_pad), generator=g) * 6.0).to(dev)
        f32_in = sp_w.key in getattr(ex.packed, "f32_in", ())
        if not f32_in and ex.split:            # a pack without the fp32 hand-over reads fp16 pairs: feed it a real split tensor instead
            xw = None
        legs = [("direct64", S["out0"], x0)] + ([("wino256", sp_w, xw)] if xw is not None else [])
        for name, sp, inp in legs:
            ex.record_variants = True
            ex.conv(sp, inp)
            ex.record_variants = False
            variant = ex.last_variant
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.15:          # ramp
                for _ in range(16):
                    ex.conv(sp, inp)
                torch.cuda.synchronize()
            pw = power_probe(lambda: [ex.conv(sp, inp) for _ in range(8)], seconds, allow_empty=True)
            if pw is None:                                  # no rocm-smi at all: plain event timing
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(256):
                    ex.conv(sp, inp)
                e1.record()
                torch.cuda.synchronize()
                pw = {"event_ms": e0.elapsed_time(e1), "loop_steps": 32, "sclk_mhz": None, "package_w": None}
            n = pw["loop_steps"] * 8
            out[name] = {"ms_per_launch": pw["event_ms"] / n, "launches": n, "sclk_mhz": pw["sclk_mhz"], "package_w": pw["package_w"], "kernel": variant,
                         "layer": "%s %d->%d, %s, 10 frames" % (sp.name, sp.cin, sp.cout, "x".join(str(d) for d in inp.shape[1:3]))}
    return out


def init_groups(dist, device, rank, world):
    """Control plane (barrier, max-over-ranks clock) on gloo; data plane (the per-layer halo slices) on RCCL
    point-to-point over xGMI.  The RCCL group is probed with one neighbour exchange before it is trusted; if any
    rank fails the probe ALL ranks fall back to host-staged halos over gloo and the JSON line says so."""
    dist.init_process_group("gloo")
    err = ""
    try:
        group = dist.new_group(backend="nccl", device_id=device)       # backend "nccl" is RCCL on ROCm
        probe_tx = torch.full((1024,), float(rank), device=device)
        probe_rx = torch.empty_like(probe_tx)
        ops = [dist.P2POp(dist.isend, probe_tx, (rank + 1) % world, group),
               dist.P2POp(dist.irecv, probe_rx, (rank - 1) % world, group)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        torch.cuda.synchronize()
        if float(probe_rx[0]) != float((rank - 1) % world):
            err = "probe payload mismatch"
    except Exception as e:                                              # noqa: BLE001 - any backend failure -> fallback
        err = "%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")
    bad = torch.tensor([1 if err else 0], dtype=torch.int32)
    dist.all_reduce(bad)                                                # gloo, CPU tensor: every rank takes the same branch
    if int(bad.item()) == 0:
        return group, "rccl point-to-point (device buffers)"
    return None, "gloo host-staged (RCCL probe failed on %d rank(s)%s)" % (int(bad.item()), "; " + err if err else "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 30 for c1, fewer for the long workloads)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--prewarm-s", type=float, default=0.5,
                    help="untimed seconds of the same step before the W warmup steps: the idle GPU sits at ~0.5 GHz and "
                         "needs a few hundred ms of load to reach its sustained clock (reported as prewarm_s)")
    ap.add_argument("--workload", default="c1", choices=sorted(WORKLOADS), help="BASELINE.json configuration (c4 = c1 with --gpus 8)")
    ap.add_argument("--frames", type=int, default=None, help="frames per GPU per step (default: the workload's)")
    ap.add_argument("--mode", default=None, choices=["clip", "stream", "perframe"],
                    help="clip: layer-major over the clip; stream: streaming_forward (rings + HIP graphs, chunked); perframe: "
                         "feedin_one_element, one graph replay per frame (default: the workload's)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --frames per GPU (the job grows with N); strong: one clip of --total-frames split over the ranks")
    ap.add_argument("--total-frames", type=int, default=80, help="--scaling strong: frames of the whole clip (C4: 80)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fuse-pairs", default="auto", choices=["auto", "on", "off"],
                    help="the 64-channel full-resolution conv pairs as one launch each (BSVD(fuse_pairs=...); auto = the product default)")
    ap.add_argument("--f32-handover", default="auto", choices=["auto", "on", "off"],
                    help="tensors only Winograd-form layers read as plain fp32 instead of fp16 pairs (BSVD(f32_handover=...); auto = on)")
    ap.add_argument("--v-handover", default="auto", choices=["auto", "on", "off"],
                    help="tensors between two F(6,3) layers in the transformed domain (BSVD(v_handover=...); auto = on where the forms allow)")
    ap.add_argument("--wide-conv", default="auto", help="arithmetic form of the wide split-fp16 layers: auto (= wino2) | direct | wino2 | wino6 | "
                                                        "wino26 (bsvd_amd.engine.WIDE_CONV; the driver's line runs the default)")
    ap.add_argument("--output-digest", action="store_true",
                    help="add `output_digest`: sha256 of every 10-frame block of the job's output (all ranks, in frame order) -- N = 1 and N = 8 "
                         "of --scaling strong run the same clip and must print the same list")
    ap.add_argument("--no-box-calibration", action="store_true", help="skip the two fixed calibration layers (~2 s) after the timed region")
    ap.add_argument("--no-power-probe", action="store_true", help="skip the 2.5 s rocm-smi power / clock sample after the timed region")
    ap.add_argument("--precision", default=DEFAULT_PRECISION, choices=["fp32", "f16x3"],
                    help="fp32: exact fp32 MFMA; f16x3: split-fp16 3-pass MFMA, fp32 accumulate (fp32-class accuracy)")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    h, w = wl["h"], wl["w"]
    mode = args.mode or wl["mode"]
    steps = args.steps if args.steps is not None else (30 if args.workload == "c1" else 5)
    warmup = args.warmup if args.warmup is not None else (5 if args.workload == "c1" else 2)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # gloo / RCCL write connection banners to fd 1 from C; keep stdout to the ONE JSON line: park fd 1 on stderr until then
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > 1 and mode != "clip":
        # the stream schedules keep per-stream state on one GPU (causality: a live stream does not shard in time);
        # running them under torch.distributed.run would time N unrelated streams under a "sharded clip" label
        raise SystemExit("--mode %s is single-GPU (replicas only); the frame-window sharded job is --mode clip" % mode)
    if args.scaling == "strong":
        if args.total_frames % world:
            raise SystemExit("--total-frames %d is not divisible by %d ranks" % (args.total_frames, world))
        frames = args.total_frames // world
    else:
        frames = args.frames if args.frames is not None else wl["frames"]
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    if os.environ.get("BSVD_BENCH_ONE_DEVICE"):     # test knob: all ranks on cuda:0 (exercises the N>1 code on a 1-GPU box)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist, halo_group, halo_transport = None, None, None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # single node: keep gloo's and RCCL's socket bootstrap on loopback (the container hostname may not resolve);
        # the halo payload itself travels over xGMI peer-to-peer, not over sockets
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        halo_group, halo_transport = init_groups(dist, device, rank, world)

    # This rank's window of the job's clip.  The clip is defined block by block (seed 100 + block index): weak scaling = one block of
    # `frames` frames per rank (rank r draws seed 100 + r, as every round's records did); strong scaling = blocks of 10 frames, so that
    # N = 1 and N = 8 run the SAME 80-frame C4 clip and their output digests can be compared block by block (`output_digest`).
    block = frames if args.scaling == "weak" else (10 if args.total_frames % 10 == 0 else args.total_frames)
    x = synth_window(rank * frames, (rank + 1) * frames, block, device, h, w, wl["blind"])      # [F,C,H,W] resident in HBM before timing

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def make_step(model, halo_fn):
        if mode == "stream":
            return lambda: model.streaming_forward(x)
        if mode == "perframe":
            S = model.shift_num

            def per_frame():       # the reference's streaming_forward loop over its own per-frame API (bsvd_arch.py:517-544)
                outs = []
                for k in range(frames + S):
                    y = model.feedin_one_element(x[k:k + 1] if k < frames else None)
                    if y is not None:
                        outs.append(y)
                model.feedin_one_element(None)
                model.reset()
                return torch.cat(outs, dim=0)
            return per_frame
        return lambda: model.clip_forward(x, halo_fn)

    def timed_run(precision, steps, warmup, prewarm_s=0.0, instrument=True, probe_s=0.0):
        """W untimed + K timed steps of the hot path at `precision`; returns (model, max-over-ranks seconds, per-kernel
        launch timings, last output)."""
        model = build_model(device, precision, wl["blind"], args.wide_conv, {"auto": "auto", "on": True, "off": False}[args.fuse_pairs],
                            {"auto": "auto", "on": True, "off": False}[args.f32_handover], {"auto": "auto", "on": True, "off": False}[args.v_handover])
        ex = model._executor(device)
        halo_fn = None
        if world > 1:
            from bsvd_amd.dist import HaloExchanger
            halo_fn = HaloExchanger(ex, rank, world, group=halo_group)
        step = make_step(model, halo_fn)
        with torch.no_grad():
            if prewarm_s > 0:                                        # clock ramp from idle, untimed
                t_pre = time.perf_counter()
                step()
                torch.cuda.synchronize()
                one = torch.tensor([time.perf_counter() - t_pre], dtype=torch.float64)
                if dist is not None:                                 # same step count on every rank (halo pairing)
                    dist.all_reduce(one, op=dist.ReduceOp.MAX)
                for _ in range(min(200, int(prewarm_s / max(float(one.item()), 1e-3)))):
                    step()
                torch.cuda.synchronize()
            timer = LaunchTimer(ex) if (instrument and mode == "clip") else None
            for _ in range(warmup):
                if timer:
                    timer.reset()
                y = step()
            if warmup and timer:
                timer.reserve(steps)
            barrier()
            if timer:
                timer.reset()
            t0 = time.perf_counter()
            for _ in range(steps):
                y = step()
            barrier()
            dt = time.perf_counter() - t0
            agg = None
            if timer:
                timer.detach()
                agg = timer.summary()
            model.bench_power = power_probe(step, probe_s) if (probe_s > 0 and world == 1) else None
            if timer:
                pass
            elif instrument:
                # stream schedules replay HIP graphs (no per-launch host call to bracket): the per-kernel table comes from one
                # extra, untimed pass of the same step with the engine issuing the same plans layer by layer
                model.bench_stream_stats = None
                if model._stream_engs:
                    e = list(model._stream_engs.values())[-1]
                    model.bench_stream_stats = dict(e.stats, chunk=e.chunk, ring_GB=round(e.ring_bytes / 1e9, 2),
                                                    graphs=sum(1 for g in e.graphs.values() if g[0]), plans=len(e.plans))
                for e in model._stream_engs.values():
                    e.layerwise = True
                timer = LaunchTimer(ex)
                step()
                timer.reset()
                step()
                torch.cuda.synchronize()
                timer.detach()
                agg = timer.summary()
                for v in agg.values():         # scale to `steps` so that roofline_of's per-step figures stay per step
                    v["ms"] *= steps
                    v["flop"] *= steps
                    v["flop_issued"] *= steps
                    v["launches"] *= steps
                    v["bytes"] *= steps
                for e in model._stream_engs.values():
                    e.layerwise = False
        assert tuple(y.shape) == (frames, 3, h, w) and (bool(torch.isfinite(y).all()) or os.environ.get("BSVD_ABL_TIMING") == "1")   # (timing-only ablation builds of tools/ab_prebuilt.sh)
        model.bench_digests = None
        if args.output_digest:
            mine_d = output_digests(y, rank * frames)
            if dist is not None:
                allg = [None] * world
                dist.all_gather_object(allg, mine_d)
                mine_d = {k: v for d_ in allg for k, v in d_.items()}
            model.bench_digests = [mine_d[k] for k in sorted(mine_d)]
        t_max = torch.tensor([dt], dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
            # self-diagnosing N > 1 line: what every rank actually exchanged (16 temporal-fusion layers per forward; interior
            # ranks send 2 slices per layer, the outermost ranks 1) and how long its own steps took
            mine = {"rank": rank, "device": torch.cuda.get_device_name(device), "exchanges": halo_fn.exchanges,
                    "bytes_sent": halo_fn.bytes_sent, "host_staged": bool(halo_fn.host_staging), "timed_s": dt,
                    "forwards": halo_fn.exchanges // 16}
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)
            model.bench_halo_stats = gathered
        return model, float(t_max.item()), agg, y

    def roofline_of(agg, precision, steps):
        peak = PEAK_F32_MFMA_TFLOPS if precision == "fp32" else PEAK_F16_MFMA_TFLOPS
        dom = max(agg, key=lambda k: agg[k]["ms"])
        ach = agg[dom]["flop"] / (agg[dom]["ms"] * 1e-3) / 1e12
        traffic, traffic_src, traffic_match = None, None, None
        try:    # HBM bytes per launch of the dominant kernel from the committed PMC passes (tools/make_traffic.py)
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            # the table is a committed file: say whether the kernel sources it was collected on are the ones this line runs
            traffic_match = (tj.get("sources_sha16") == kernel_sources_sha16()) if tj.get("sources_sha16") else None
            if args.workload == "c1" and mode == "clip" and frames == 10:      # the table was collected on this workload
                traffic = tj["kernels"][dom]["hbm_bytes_per_launch"]
                traffic_src = "profiles/traffic.json (%s; %s)" % (tj["source"], tj["formula"])
        except (OSError, KeyError, ValueError):
            pass
        traffic_alg = agg[dom]["bytes"] / agg[dom]["launches"]
        return {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC, separate passes)",
                "traffic_source": traffic_src, "traffic_sources_match": traffic_match,
                # the least a launch of this kernel has to move (inputs, outputs, skip / residual operand and weights once each, averaged
                # over the kernel's launches of this step) and what it moves relative to that: > 1 = re-reads
                "traffic_algorithmic": traffic_alg, "traffic_ratio": (traffic / traffic_alg) if traffic else None,
                "hbm_gbps_algorithmic": traffic_alg / (agg[dom]["ms"] / agg[dom]["launches"] * 1e-3) / 1e9,
                "timing": "HIP events around every launch inside the timed region" if mode == "clip" else
                          "HIP events around every launch in one extra untimed pass of the same step plans issued layer by layer "
                          "(the timed region replays them as HIP graphs)",
                "note": ("algorithmic FLOP over the fp32-MFMA peak (v_mfma_f32_32x32x2_f32)" if precision == "fp32" else
                         "algorithmic FLOP over the dense fp16-MFMA peak; the 3-pass split issues 3 MFMA FLOP per "
                         "algorithmic FLOP (a Winograd F(m,3) kernel 3 (m + 2) / m: mfma_flop_per_algorithmic_flop), so MFMA-pipe "
                         "utilisation = that factor x frac (mfma_pipe_frac).  Whether THIS run sat at the package power limit is "
                         "measured, not assumed: power.at_power_cap (package_w >= power.at_power_cap_threshold x cap_w); for scale, "
                         "the guide's own dense bf16 GEMM on random data sustains 1,247 TFLOP/s = 0.50 of the 2.5 PFLOP/s peak at "
                         "that limit (MI355X_MICROARCH.md, DVFS give-back)"),
                "mfma_flop_per_algorithmic_flop": agg[dom]["flop_issued"] / agg[dom]["flop"],
                "mfma_flop_issued_per_launch": agg[dom]["flop_issued"] / agg[dom]["launches"],
                "mfma_pipe_frac": agg[dom]["flop_issued"] / (agg[dom]["ms"] * 1e-3) / 1e12 / peak,
                "avg_launch_ms": agg[dom]["ms"] / agg[dom]["launches"], "launches": agg[dom]["launches"],
                "all_conv_kernels": {k: {"ms_per_step": v["ms"] / steps, "tflops": v["flop"] / (v["ms"] * 1e-3) / 1e12,
                                         "mfma_tflops_issued": v["flop_issued"] / (v["ms"] * 1e-3) / 1e12,
                                         "launches_per_step": v["launches"] // steps} for k, v in agg.items()},
                "conv_ms_per_step": sum(v["ms"] for v in agg.values()) / steps}

    # ---- the timed job (headline) ...
    model, elapsed, agg, y = timed_run(args.precision, steps, warmup, args.prewarm_s, probe_s=0.0 if args.no_power_probe else 5.0)
    stream_stats = getattr(model, "bench_stream_stats", None)
    model.release_stream_buffers()
    # ---- ... and, outside it, the other arithmetic mode on the same clip for reference + a live parity figure
    other = "fp32" if args.precision == "f16x3" else "f16x3"
    steps_o = max(1, min(steps, 3))
    model_o, elapsed_o, agg_o, y_o = timed_run(other, steps_o, 1)
    model_o.release_stream_buffers()
    parity = float((y.float() - y_o.float()).abs().max())

    if rank == 0:
        total_frames = frames * world * steps
        fps = total_frames / elapsed
        flop_per_frame = 2.0 * model.net.macs_per_frame(h, w)
        degraded = bool(halo_transport) and not halo_transport.startswith("rccl")
        api = {"clip": "BSVD.clip_forward on the pre-concatenated [F,4,H,W] clip = BSVD.forward minus the 5-D reshape and the noise-map "
                       "torch.cat (clip schedule: layer-major, every launch of the net inside the timed region)",
               "stream": "BSVD.streaming_forward (stream schedule on rings, HIP-graph replay, %s frames per pipeline step)"
                         % (stream_stats["chunk"] if stream_stats else "?"),
               "perframe": "BSVD.feedin_one_element per frame + 17 flush feeds (one HIP-graph replay per frame)"}[mode]
        out = {
            "metric": "denoised frames/sec @%dx%d sigma=30 (bsvd_c64 streaming bidirectional-buffer forward)" % (h, w),
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": warmup, "prewarm_s": args.prewarm_s,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            # the reference's own published 28.3 frames/s for BSVD.forward on this very clip shape (one
            # unnamed CUDA GPU, fp16 weights + autocast) -- a single-GPU number, so the ratio is reported at N = 1 only
            "vs_baseline": (fps / REFERENCE_PUBLISHED_FPS) if (world == 1 and args.workload == "c1") else None,
            "vs_baseline_source": "BASELINE.md s1: 0.353594 s per [1,10,4,540,960] clip = 28.3 frames/s (reference README.md:88-106; "
                                  "other hardware, fp16 autocast)",
            "dtype": "f32" if args.precision == "fp32" else "f16x3 (split-fp16 MFMA, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": (wl["name"] % (frames * world)) + ", %s, random-init weights%s"
                                   % (api, "; frame-window sharded with per-layer RCCL halo" if world > 1 else ""),
                       "baseline_config": ("c4" if frames * world == 80 else args.workload if world == 1 else "c1 x%d" % world) if args.workload == "c1" else args.workload,
                       "frames_per_gpu": frames, "parallelism": "frame-window x%d" % world,
                       "schedule": mode, "halo_transport": halo_transport, "wide_conv": model.wide_conv, "fuse_pairs": model.fuse_pairs,
                       "v_handover_layers": len(getattr(model._packed, "v_out", ())),
                       "flop_per_frame": flop_per_frame},
            # true when the halo slices of an N>1 run did NOT travel over RCCL/xGMI (host-staged gloo fallback): such a line is a
            # functional check, not a scaling measurement
            "degraded": degraded,
            "path_tflops": fps * flop_per_frame / 1e12,
            "path_frac_of_mfma_peak": fps * flop_per_frame / 1e12 /
                                      ((PEAK_F32_MFMA_TFLOPS if args.precision == "fp32" else PEAK_F16_MFMA_TFLOPS) * world),
            "roofline": roofline_of(agg, args.precision, steps),
            "parity": {"max_abs_f16x3_vs_exact_fp32_on_this_clip": parity, "budget": 1e-3,
                       "note": "north_star: <= 1e-3 max-abs vs the fp32 forward; tests/test_gpu_f16x3.py pins 3-6e-5 vs "
                               "the reference goldens"},
            "other_mode": {"dtype": "f32" if other == "fp32" else "f16x3", "value": frames * world * steps_o / elapsed_o,
                           "unit": "frames/s", "steps": steps_o, "ms_per_step": elapsed_o / steps_o * 1e3,
                           "roofline": roofline_of(agg_o, other, steps_o)},
        }
        pw = getattr(model, "bench_power", None)
        if pw:
            # what the dominant kernel's MFMA work is against the matrix-pipe peak AT THE CLOCK THE CHIP ACTUALLY RUNS (2.4 GHz nominal)
            scale = pw["sclk_mhz"] / 2400.0
            # ONE threshold, stated: rocm-smi's package power is an average over its sampling window and the limiter holds the chip a
            # few percent under the cap (r04: 1359-1384 W of 1400 at 1.85-1.90 GHz of 2.4), so "at the cap" = within 5 % of it AND
            # the shader clock pulled below nominal
            pw["at_power_cap_threshold"] = 0.95
            pw["frac_of_cap"] = (pw["package_w"] / pw["cap_w"]) if pw["cap_w"] else None
            pw["at_power_cap"] = bool(pw["cap_w"]) and pw["package_w"] >= 0.95 * pw["cap_w"] and pw["sclk_mhz"] < 0.95 * 2400.0
            pw["mfma_pipe_frac_at_sampled_clock"] = out["roofline"]["mfma_pipe_frac"] / scale if scale > 0 else None
            out["power"] = pw
            # the timed region is a short burst (the reference's protocol: profile.py takes the best of 10 short runs); a live stream
            # gets what the chip sustains at its package power cap -- the same step looped for >= 5 s right after the timed region
            out["sustained"] = {"value": frames * pw["loop_steps"] / pw["loop_seconds"], "unit": "frames/s", "seconds": pw["loop_seconds"],
                                "sclk_mhz": pw["sclk_mhz"], "package_w": pw["package_w"],
                                "note": "untimed loop of the same step after the timed region (N = 1); `value` above is the K-step burst"}
        if world == 1 and not args.no_box_calibration and not wl["blind"]:
            try:
                cal = box_calibration(model, x)
            except Exception as e:                                   # noqa: BLE001 - a measurement aid must never fail the bench
                cal = {"error": "%s: %s" % (type(e).__name__, e)}
            cal["reference"] = BOX_CAL_REF
            d64 = cal.get("direct64", {}).get("ms_per_launch")
            cal["ratio_direct64"] = (d64 / BOX_CAL_REF["direct64_ms"]) if d64 else None
            cal["note"] = ("fixed layers on seeded operands, looped ~0.7 s each after the timed region; value_normalised = value x "
                           "(this box's direct64 ms / the reference box's): the frames/s this commit would show on the reference box")
            out["box_calibration"] = cal
            out["value_normalised"] = normalise_value(fps, d64, BOX_CAL_REF["direct64_ms"])
        if getattr(model, "bench_digests", None):
            out["output_digest"] = {"sha256_16_per_10_frame_block": model.bench_digests, "blocks": len(model.bench_digests),
                                    "of": "the last timed step's output, frame order over all ranks"}
        if stream_stats:
            out["stream_engine"] = stream_stats
        if world > 1:
            out["halo"] = {"transport": halo_transport, "degraded": degraded,
                           "expected": "16 exchanges per forward and rank; per forward an interior rank sends 2 x 99.5 MB (fp32 / "
                                       "split16 alike), the first and last rank 1 x 99.5 MB at 540x960",
                           "per_rank": getattr(model, "bench_halo_stats", None)}
        if world == 1 and not args.no_cpu_baseline:
            # bounded sample of the same workload: 10 frames of its geometry (4 at 1080p), the CPU path's cost is linear in frames
            out["cpu_baseline"] = cpu_baseline(model, h, w, wl["blind"], frames=4 if h * w > 540 * 960 else 10)
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
