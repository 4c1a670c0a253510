#!/bin/bash
# A/B of compile-time tuning variants ON THE GPU BOX, interleaved in one session (box-to-box variance is large).
# usage: tools/ab.sh "<flags A>" "<flags B>" ...   e.g.  tools/ab.sh "" "-DBSVD_TUNE_D=2" "-DBSVD_TUNE_ALIGN=1"
cd $GRAFT_REPO_ROOT
i=0
for f in "$@"; do
  EXTRA_HIPCC_FLAGS="$f" BSVD_OBJ_SUFFIX=_ab$i BSVD_OUT=/tmp/libbsvd_ab$i.so bsvd_amd/csrc/build.sh > /dev/null 2>&1 || echo "build $i failed"
  i=$((i+1))
done
for round in 1 2; do
  i=0
  for f in "$@"; do
    echo -n "[$round] '$f': "
    BSVD_HIP_LIB=/tmp/libbsvd_ab$i.so python bench.py --no-cpu-baseline ${AB_BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('fps %.1f other %.1f' % (d['value'], d['other_mode']['value']), {k.replace('conv3x3_kernel',''):round(v['ms_per_step'],2) for k,v in r['all_conv_kernels'].items()}, {k.replace('conv3x3_kernel',''):round(v['ms_per_step'],2) for k,v in d['other_mode']['roofline']['all_conv_kernels'].items()})"
    i=$((i+1))
  done
done
