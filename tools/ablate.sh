#!/bin/bash
# Timing-only ablation of the conv kernel's main loop (run ON THE GPU BOX; rebuilds the box's copy of the
# library with -DBSVD_ABLATE).  bit0: no weight loads, bit1: no patch prefetch, bit2: no A-fragment LDS reads.
cd $GRAFT_REPO_ROOT
touch bsvd_amd/csrc/*.hip
EXTRA_HIPCC_FLAGS=-DBSVD_ABLATE bsvd_amd/csrc/build.sh > /dev/null
for m in ${ABLATE_MODES:-0 1 2 3 4 7}; do
  echo "== BSVD_ABLATE=$m"
  BSVD_ABLATE=$m python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('fps %.1f  step %.1f ms' % (d['value'], d['ms_per_step']), {k:(round(v['ms_per_step'],1), round(v['tflops'],1)) for k,v in r['all_conv_kernels'].items()})"
done
