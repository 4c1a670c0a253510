#!/bin/bash
# Round-5 GPU session 12 (generic A/B of the Winograd kernel variants in build/ab/): digests, layer loops, whole C1 clip interleaved.
# digests, layer loops, whole C1 clip interleaved.     needs: tools/build_ab.sh "" "-DBSVD_WX_ILV=3" [more variants]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
f() { grep -v "amdgpu.ids\|BSVD_HIP_LIB"; }
N=$(wc -l < build/ab/variants.txt)
{ cat build/ab/variants.txt
  for i in $(seq 0 $((N-1))); do echo "== ab$i"; BSVD_HIP_LIB=$R/build/ab/lib_ab$i.so timeout 600 python tests/measure_driver.py digest $O/r05n_digest_ab$i.json 2>&1 | f | tail -1
    [ $i -gt 0 ] && { cmp $O/r05n_digest_ab0.json $O/r05n_digest_ab$i.json && echo "digest ab$i == ab0"; }
    BSVD_HIP_LIB=$R/build/ab/lib_ab$i.so timeout 600 python tools/debug/wino_f32_bench.py 2 wino2 2>&1 | f | grep "in f32   out f32"; done
  for round in 1 2 3; do for i in $(seq 0 $((N-1))); do
    echo -n "[$round] ab$i: "
    BSVD_HIP_LIB=$R/build/ab/lib_ab$i.so timeout 300 python bench.py --no-cpu-baseline --no-power-probe --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('fps %.1f parity %.3e' % (d['value'], d['parity']['max_abs_f16x3_vs_exact_fp32_on_this_clip']), {k.replace('conv3x3_kernel','').replace('winox_kernel','wx'):round(v['ms_per_step'],3) for k,v in r['all_conv_kernels'].items()})"
  done; done; } > $O/r05n_ab.txt 2>&1
cat $O/r05n_ab.txt
