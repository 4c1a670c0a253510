#!/bin/bash
# A/B of PREBUILT variant libraries (build/ab/lib_ab<i>.so, built in the container with EXTRA_HIPCC_FLAGS),
# interleaved on one GPU box.  usage: tools/ab_prebuilt.sh <n variants> [rounds]
cd $GRAFT_REPO_ROOT
N=${1:-2}; R=${2:-2}
for round in $(seq 1 $R); do
  for i in $(seq 0 $((N-1))); do
    echo -n "[$round] ab$i: "
    BSVD_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/lib_ab$i.so python bench.py --no-cpu-baseline --steps 20 --warmup 3 ${AB_BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('fps %.1f other %.1f parity %.2e' % (d['value'], d['other_mode']['value'], d['parity']['max_abs_f16x3_vs_exact_fp32_on_this_clip']), {k.replace('conv3x3_kernel',''):round(v['ms_per_step'],2) for k,v in r['all_conv_kernels'].items()}, {k.replace('conv3x3_kernel',''):round(v['ms_per_step'],2) for k,v in d['other_mode']['roofline']['all_conv_kernels'].items()})"
  done
done
