#!/bin/bash
# Samples rocm-smi (power, clocks, temperature) while the bench loop runs -- is the split-fp16 mode power-capped?
cd $GRAFT_REPO_ROOT
rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor junction\)" | head -8
python - <<'PY' &
import sys, time, torch
sys.path.insert(0, '.')
import bench
m = bench.build_model(torch.device('cuda', 0), sys.argv[1] if len(sys.argv) > 1 else 'f16x3')
lq, nm = bench.synth_clip(10, 100, 'cuda')
x = torch.cat([lq, nm], dim=2)[0].contiguous()
t0 = time.time()
n = 0
while time.time() - t0 < 12:
    for _ in range(10):
        m.clip_forward(x)
    torch.cuda.synchronize(); n += 10
print("steps", n, "fps", n * 10 / (time.time() - t0))
PY
sleep 5
for i in 1 2 3; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|Temperature \(Sensor junction\)" | tr '\n' ' '; echo; sleep 2; done
wait
