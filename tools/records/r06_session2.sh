#!/bin/bash
# Round 6, GPU session 2: consumer ceiling of the transformed-domain hand-over: every Winograd form with its in-kernel transform
# vs a transformed-domain input (tools/debug/v_layer_bench.py), and the whole-clip error of F(6,3) against the CPU oracle.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s2; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python tools/debug/v_layer_bench.py 1.5 wino2,wino6 > $O/v_layers.txt 2> $O/v_layers.err; echo "rc $?"; cat $O/v_layers.txt; tail -3 $O/v_layers.err
BSVD_HIP_LIB=$GRAFT_REPO_ROOT/build/measure/libbsvd_hip.so V_LAYERS=0,1,2,3 timeout 600 python tools/debug/v_layer_bench.py 1.5 wino4 > $O/v_layers_wino4.txt 2> $O/v_layers_wino4.err; echo "rc $?"; cat $O/v_layers_wino4.txt; tail -3 $O/v_layers_wino4.err
timeout 900 python tools/full_clip_parity.py --workloads c1,c2,c3 --frames 10,10,10 --wide-conv wino6 --json $O/full_clip_parity_wino6.json > $O/parity_wino6.txt 2>&1; echo "rc $?"
python - <<'PY'
import json
for r in json.load(open("gpurun_out/r06_s2/full_clip_parity_wino6.json"))["rows"]:
    print(r["workload"], r["wide_conv"], r["clip"], "f16x3 max-abs", r["f16x3"]["max_abs_vs_oracle"], "fp32", r["fp32"]["max_abs_vs_oracle"], "out max", r["output_max"])
PY
