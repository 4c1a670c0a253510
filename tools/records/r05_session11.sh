#!/bin/bash
# Round-5 GPU session 11: wave priority in the Winograd K loop (-DBSVD_WX_SPRIO=0|1|2|3), bit-identical by construction: whole C1 clip interleaved.
# needs: tools/build_ab.sh "" "-DBSVD_WX_SPRIO=1" "-DBSVD_WX_SPRIO=2" "-DBSVD_WX_SPRIO=3"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
{ cat build/ab/variants.txt
  for round in 1 2 3; do for i in 0 1 2 3; do
    echo -n "[$round] ab$i: "
    BSVD_HIP_LIB=$R/build/ab/lib_ab$i.so python bench.py --no-cpu-baseline --no-power-probe --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('fps %.1f parity %.3e' % (d['value'], d['parity']['max_abs_f16x3_vs_exact_fp32_on_this_clip']), {k.replace('conv3x3_kernel','').replace('winox_kernel','wx'):round(v['ms_per_step'],3) for k,v in r['all_conv_kernels'].items()})"
  done; done; } > $O/r05l_sprio.txt 2>&1
cat $O/r05l_sprio.txt
