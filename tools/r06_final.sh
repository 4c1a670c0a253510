#!/bin/bash
# The round's final records in ONE box session: the whole GPU suite + every profiles/r06_* record (tools/round_profiles.sh), then the C1 line
# re-taken against the traffic table of this very session (bench.py reads profiles/traffic.json) with its rocprofv3 kernel stats, the C4 clip at
# N = 1 and through 8 host-staged ranks on one device, and the single-frame layer table.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
export PYTHONDONTWRITEBYTECODE=1
bash tools/round_profiles.sh r06
cp $O/r06_traffic.json $R/profiles/traffic.json
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/r06_c1.json 2> $O/r06_c1.err; tail -c 400 $O/r06_c1.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r06_c1 -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-power-probe --no-box-calibration > /dev/null 2> $O/prof_r06_c1.log
f=$(find $O/prof_r06_c1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r06_c1_kernel_stats.csv; rm -rf $O/prof_r06_c1
grep -i "winox" $O/r06_c1_kernel_stats.csv | head -3
cd $R
C="--steps 3 --warmup 1 --no-cpu-baseline --scaling strong --total-frames 80 --output-digest"
python bench.py --gpus 1 $C > $O/r06_c4_n1.json 2> $O/r06_c4_n1.err
BSVD_BENCH_ONE_DEVICE=1 MASTER_ADDR=127.0.0.1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 $C --no-power-probe > $O/r06_c4_n8_one_device.json 2> $O/r06_c4_n8.err
python - <<'PY'
import json, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out"
for f in ("r06_c4_n1","r06_c4_n8_one_device"):
    try:
        d=json.loads([l for l in open(O+"/%s.json"%f) if l.startswith("{")][0]); print(f, round(d["value"],1), d["degraded"], d["output_digest"]["sha256_16_per_10_frame_block"][:2])
    except Exception as e: print(f, "failed", e)
PY
python tools/per_layer_stream.py 2>&1 | grep -v "amdgpu.ids" > $O/r06_per_layer_stream.txt; tail -1 $O/r06_per_layer_stream.txt
