#!/bin/bash
# Round 6, GPU session 9: the C2 clip (85 x 480 x 856) through the r05 tree, the hygiene commit and the current tree, interleaved -- the final-record session
# showed its Winograd launches 16 % slower than r05's record while C1 / C3 / C5 were not.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_s9; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
for r in 1 2; do for t in build/wt_r05 build/wt_hyg .; do
  echo -n "[$r] $t: "; (cd $t && python bench.py --workload c2 --no-cpu-baseline --steps 3 --warmup 1 --no-power-probe 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('fps %.1f' % d['value'], {k.replace('conv3x3_kernel','').replace('winox_kernel',''):round(v['ms_per_step'],1) for k,v in r['all_conv_kernels'].items()})")
done; done 2>&1 | tee $O/c2_bisect.txt
