#!/bin/bash
# Round-5 GPU session 5: the plain-fp32 hand-over between Winograd-form layers (BsvdConvArgs.y_f32 / x_f32): parity, then interleaved A/B.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
f() { grep -v amdgpu.ids; }
timeout 900 python -m pytest tests/test_gpu_f32_handover.py tests/test_gpu_wino.py tests/test_gpu_stream_graph.py tests/test_gpu_stress.py -q -x 2>&1 | f | tail -30 > $O/r05d_handover_tests.txt
tail -12 $O/r05d_handover_tests.txt
{ for round in 1 2 3; do for h in off on; do
    echo -n "[$round] f32_handover=$h: "
    python bench.py --f32-handover $h --no-cpu-baseline --no-power-probe --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('fps %.1f parity %.2e' % (d['value'], d['parity']['max_abs_f16x3_vs_exact_fp32_on_this_clip']), {k.replace('conv3x3_kernel','').replace('winox_kernel','wx'):(round(v['ms_per_step'],3), v['launches_per_step']) for k,v in r['all_conv_kernels'].items()})"
  done; done; } > $O/r05d_handover_ab.txt 2>&1
cat $O/r05d_handover_ab.txt
