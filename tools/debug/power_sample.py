"""Clocks / package power (rocm-smi) while the C1 clip loop runs: is a precision mode running at the power limit?
usage (GPU box): python tools/debug/power_sample.py [f16x3|fp32] [seconds=8]"""
import os, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
dev = torch.device("cuda", 0)
m = bench.build_model(dev, precision=prec)
x = torch.rand((1, 10, 4, 540, 960), device=dev)
for _ in range(3):
    m(x)
torch.cuda.synchronize()
stop = False
rows = []


def sample():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True).stdout
        keep = [l.split(":", 1)[1].strip() if ":" in l else l for l in out.splitlines()
                if any(k in l.lower() for k in ("sclk", "mclk", "power (w)", "junction", "socket power"))]
        rows.append((time.time(), keep))
        time.sleep(0.7)


th = threading.Thread(target=sample)
th.start()
t0 = time.time()
n = 0
while time.time() - t0 < secs:
    for _ in range(5):
        m(x)
    torch.cuda.synchronize()
    n += 5
el = time.time() - t0
stop = True
th.join()
print("%s: %.1f frames/s over %.1f s" % (prec, n * 10 / el, el))
for t, keep in rows:
    print("  t=%.1f " % (t - t0), " | ".join(keep))
print(subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True).stdout)
