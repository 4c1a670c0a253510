#!/bin/bash
# Round 6, GPU session 3: the transformed-domain hand-over end to end -- its tests, the producer's cost per layer, and the C1 clip with
# F(2,3) (shipped) / F(6,3) / F(6,3) + hand-over interleaved on one box.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s3; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_v_handover.py -x -q > $O/pytest_v.log 2>&1; echo "pytest rc $?"; tail -25 $O/pytest_v.log
V_LAYERS=0,1,4,5 timeout 600 python tools/debug/v_layer_bench.py 1.5 wino6 > $O/v_layers_out.txt 2> $O/v_layers_out.err; echo "rc $?"; cat $O/v_layers_out.txt; tail -3 $O/v_layers_out.err
for round in 1 2; do
  for cfg in "wino2 auto" "wino6 off" "wino6 on"; do
    set -- $cfg
    echo -n "[$round] $1 v=$2: "
    timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 --no-power-probe --no-box-calibration --wide-conv $1 --v-handover $2 2>$O/bench_$1_$2.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('fps %.1f parity %.2e vlayers %s' % (d['value'], d['parity']['max_abs_f16x3_vs_exact_fp32_on_this_clip'], d['config'].get('v_handover_layers')), {k.replace('conv3x3_kernel','').replace('winox_kernel',''):round(v['ms_per_step'],2) for k,v in r['all_conv_kernels'].items()})" || tail -5 $O/bench_$1_$2.err
  done
done 2>&1 | tee $O/ab_forms.txt
