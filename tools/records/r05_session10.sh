#!/bin/bash
# Round-5 GPU session 10: the short last row band of 16-row Winograd tile grids as FOLDED tiles (-DBSVD_WX_TAIL=2, the default) against the
# 8-row body (=1): tests on the default build, output digests of both builds, stream schedules (the per-frame API is what it is for),
# the single-frame layer table, C1 / C2 clips interleaved.      needs: tools/build_ab.sh "-DBSVD_WX_TAIL=1" ""
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
f() { grep -v "amdgpu.ids\|BSVD_HIP_LIB"; }
timeout 1200 python -m pytest tests/test_gpu_wino.py tests/test_gpu_f32_handover.py tests/test_gpu_fullsize.py tests/test_gpu_stream_graph.py tests/test_gpu_fuzz.py -q -x 2>&1 | f | tail -8
{ for i in 0 1; do echo "== ab$i $(sed -n "$((i+1))p" build/ab/variants.txt)"; BSVD_HIP_LIB=$R/build/ab/lib_ab$i.so python tests/measure_driver.py digest $O/r05k_digest_ab$i.json 2>&1 | f | tail -1; done
  cmp $O/r05k_digest_ab0.json $O/r05k_digest_ab1.json && echo "digests of the two builds are identical"
  for round in 1 2; do for i in 0 1; do
    echo "[$round] ab$i stream schedules, 540x960:"
    BSVD_HIP_LIB=$R/build/ab/lib_ab$i.so python tools/stream_modes.py --frames 85 --reps 3 2>&1 | f | tail -8
  done; done
  for i in 0 1; do echo "== ab$i single-frame launches"; BSVD_HIP_LIB=$R/build/ab/lib_ab$i.so python tools/per_layer_stream.py 2>&1 | f | grep "downc1.memconv\|upc2\|sum:"; done
  for round in 1 2 3; do for i in 0 1; do
    echo -n "[$round] ab$i: "
    BSVD_HIP_LIB=$R/build/ab/lib_ab$i.so python bench.py --no-cpu-baseline --no-power-probe --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('fps %.1f parity %.3e' % (d['value'], d['parity']['max_abs_f16x3_vs_exact_fp32_on_this_clip']), {k.replace('conv3x3_kernel','').replace('winox_kernel','wx'):round(v['ms_per_step'],3) for k,v in r['all_conv_kernels'].items()})"
  done; done
  for i in 0 1; do echo -n "c2 ab$i: "; BSVD_HIP_LIB=$R/build/ab/lib_ab$i.so python bench.py --workload c2 --no-cpu-baseline --no-power-probe 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fps %.1f' % d['value'])"; done; } > $O/r05k_fold.txt 2>&1
cat $O/r05k_fold.txt
