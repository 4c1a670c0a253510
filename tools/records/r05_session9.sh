#!/bin/bash
# Round-5 GPU session 9: the 8-row body for the short last row band of 16-row tile grids (winox_kernel_tail; -DBSVD_WX_TAIL=0 | 1).
# needs: tools/build_ab.sh "-DBSVD_WX_TAIL=0" ""
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
f() { grep -v "amdgpu.ids\|BSVD_HIP_LIB"; }
timeout 900 python -m pytest tests/test_gpu_wino.py tests/test_gpu_f32_handover.py tests/test_gpu_fullsize.py tests/test_gpu_stream_graph.py -q -x 2>&1 | f | tail -6
{ for i in 0 1; do echo "== ab$i $(sed -n "$((i+1))p" build/ab/variants.txt)"; BSVD_HIP_LIB=$R/build/ab/lib_ab$i.so python tools/debug/wino_f32_bench.py 2 wino2 2>&1 | f | grep "in f32   out f32"; BSVD_HIP_LIB=$R/build/ab/lib_ab$i.so python tests/measure_driver.py digest $O/r05j_digest_ab$i.json 2>&1 | f | tail -1; done
  cmp $O/r05j_digest_ab0.json $O/r05j_digest_ab1.json && echo "digests of the two builds are identical"
  for round in 1 2 3; do for i in 0 1; do
    echo -n "[$round] ab$i: "
    BSVD_HIP_LIB=$R/build/ab/lib_ab$i.so python bench.py --no-cpu-baseline --no-power-probe --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('fps %.1f parity %.3e' % (d['value'], d['parity']['max_abs_f16x3_vs_exact_fp32_on_this_clip']), {k.replace('conv3x3_kernel','').replace('winox_kernel','wx'):round(v['ms_per_step'],3) for k,v in r['all_conv_kernels'].items()})"
  done; done
  for i in 0 1; do echo -n "c2 ab$i: "; BSVD_HIP_LIB=$R/build/ab/lib_ab$i.so python bench.py --workload c2 --no-cpu-baseline --no-power-probe 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fps %.1f' % d['value'])"; done; } > $O/r05j_tail.txt 2>&1
cat $O/r05j_tail.txt
