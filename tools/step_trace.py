"""Per-step GPU time of consecutive clip steps (HIP events, no host sync inside) -- shows the clock transient after idle.
usage: python tools/step_trace.py [steps] [idle_ms]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    idle_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    dev = torch.device("cuda", 0)
    model = bench.build_model(dev, "f16x3")
    lq, nm = bench.synth_clip(10, 100, dev)
    x = torch.cat([lq, nm], dim=2)[0].contiguous()
    with torch.no_grad():
        for phase in ("cold", "after-prewarm"):
            if phase == "after-prewarm" and idle_ms:
                torch.cuda.synchronize(); time.sleep(idle_ms * 1e-3)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
            ev[0].record()
            for i in range(n):
                model.clip_forward(x, None)
                ev[i + 1].record()
            torch.cuda.synchronize()
            ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
            print(phase, " ".join("%.1f" % m for m in ms))

if __name__ == "__main__":
    main()
