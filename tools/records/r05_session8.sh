#!/bin/bash
# Round-5 GPU session 8: how a chunk's transform items are dealt to the Winograd workgroup's waves (BSVD_WX_ROTA 0 | 1 | 2), bit-identical by construction:
# layer loops, output digests, whole C1 clip interleaved.   needs: tools/build_ab.sh "" "-DBSVD_WX_ROTA=1" "-DBSVD_WX_ROTA=2"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
f() { grep -v "amdgpu.ids\|BSVD_HIP_LIB"; }
{ for i in 0 1 2; do echo "== ab$i $(sed -n "$((i+1))p" build/ab/variants.txt)"; BSVD_HIP_LIB=$R/build/ab/lib_ab$i.so python tools/debug/wino_f32_bench.py 2 wino2 2>&1 | f | grep "in f32   out f32\|in pairs out pairs"; BSVD_HIP_LIB=$R/build/ab/lib_ab$i.so python tests/measure_driver.py digest $O/r05h_digest_ab$i.json 2>&1 | f | tail -1; done
  cmp $O/r05h_digest_ab0.json $O/r05h_digest_ab1.json && cmp $O/r05h_digest_ab0.json $O/r05h_digest_ab2.json && echo "digests of the three builds are identical"
  for round in 1 2 3; do for i in 0 1 2; do
    echo -n "[$round] ab$i: "
    BSVD_HIP_LIB=$R/build/ab/lib_ab$i.so python bench.py --no-cpu-baseline --no-power-probe --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('fps %.1f parity %.3e' % (d['value'], d['parity']['max_abs_f16x3_vs_exact_fp32_on_this_clip']), {k.replace('conv3x3_kernel','').replace('winox_kernel','wx'):round(v['ms_per_step'],3) for k,v in r['all_conv_kernels'].items()})"
  done; done; } > $O/r05h_rota.txt 2>&1
cat $O/r05h_rota.txt
