#!/bin/bash
# A/B of the FAST vs GENERIC main loop of the conv kernel ON THE GPU BOX (a -DBSVD_ABLATE build reads BSVD_ABLATE=8 as
# "force the generic path").  Finer ablations (dropping the weight / slice / A-fragment loads behind runtime flags)
# were used in round 1 on the first kernels (102 -> 130 TFLOP/s with all loads removed, which motivated the 2-step
# register rings); on the current branch-free loop such flags perturb the schedule more than they reveal.
cd $GRAFT_REPO_ROOT
EXTRA_HIPCC_FLAGS=-DBSVD_ABLATE BSVD_OBJ_SUFFIX=_abl BSVD_OUT=/tmp/libbsvd_abl.so bsvd_amd/csrc/build.sh > /dev/null
for m in 0 8; do
  echo "== BSVD_ABLATE=$m"
  BSVD_HIP_LIB=/tmp/libbsvd_abl.so BSVD_ABLATE=$m python bench.py --steps 3 --warmup 1 --no-cpu-baseline --precision fp32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('fps %.1f  step %.1f ms' % (d['value'], d['ms_per_step']), {k.replace('conv3x3_kernel',''):(round(v['ms_per_step'],1), round(v['tflops'],1)) for k,v in r['all_conv_kernels'].items()})"
done
