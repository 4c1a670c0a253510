"""Probe of the GPU box's usable host cores for the cpu_baseline leg (tools only)."""
import os, time, torch, torch.nn.functional as F
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(p, open(p).read().strip())
    except OSError: pass
x = torch.randn(1, 64, 540, 960); w = torch.randn(64, 64, 3, 3)
for nt in (8, 16, 32, 64, 128, 256):
    torch.set_num_threads(nt)
    F.conv2d(x, w, padding=1)
    t0 = time.perf_counter()
    for _ in range(3): F.conv2d(x, w, padding=1)
    dt = (time.perf_counter() - t0) / 3
    print("threads %3d: %.1f ms/conv  %.0f GFLOP/s" % (nt, dt * 1e3, 2 * 64 * 64 * 9 * 540 * 960 / dt / 1e9))
