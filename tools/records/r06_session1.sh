#!/bin/bash
# Round 6, GPU session 1: the cleaned sources (knob hygiene) against the r05 sources on one box, the GPU suite, the bench line with
# box_calibration, and the C4 clip at N = 1 / through 8 host-staged ranks on one device.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s1; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc; tail -5 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06_s1/bench.json"))
print("fps", d["value"], "normalised", d.get("value_normalised"), "frac", d["roofline"]["frac"]); print(json.dumps(d.get("box_calibration"), indent=1)[:1500])
PY
timeout 900 tools/ab_prebuilt.sh 2 2 > $O/ab_clean_vs_r05.txt 2>&1; cat $O/ab_clean_vs_r05.txt
C="--steps 3 --warmup 1 --no-cpu-baseline --scaling strong --total-frames 80 --output-digest"
timeout 900 python bench.py --gpus 1 $C > $O/c4_n1.json 2> $O/c4_n1.err; echo "c4 n1 rc $?"
BSVD_BENCH_ONE_DEVICE=1 MASTER_ADDR=127.0.0.1 timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 $C --no-power-probe > $O/c4_n8_one_device.json 2> $O/c4_n8.err; echo "c4 n8 rc $?"
python - <<'PY'
import json
for f in ("c4_n1","c4_n8_one_device"):
    try:
        d=json.loads([l for l in open("gpurun_out/r06_s1/%s.json"%f) if l.startswith("{")][0])
        print(f, d["value"], d["ms_per_step"], d["degraded"], d["output_digest"]["sha256_16_per_10_frame_block"])
    except Exception as e: print(f, "failed", e)
PY
