#!/bin/bash
# The Winograd kernel's measurement record in one GPU-box session (profiles/<tag>_wino_*).
# Needs the variant libraries of
#   tools/build_ab.sh "" "-DBSVD_WX_ABL=1" "-DBSVD_WX_ABL=2" "-DBSVD_WX_ABL=3" "-DBSVD_WX_ABL=4" "-DBSVD_WX_ABL=8" "-DBSVD_WX_ABL=64" "-DBSVD_WX_EMAP=0 -DBSVD_WX_XIN=0" "-DBSVD_WX_TL"
# (ab0 shipped; 1 no transform; 2 no MFMA steps; 3 neither; 4 no epilogue finish; 8 no chunk barrier; 64 no activation loads in the K loop;
#  ab7 the round's first load path: quarter-major items, compare / select per position; ab8 timeline)
# usage on the GPU box: tools/wino_records.sh <tag>
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
f() { grep -v amdgpu.ids; }
# 1. every form, layer by layer, on the same realistic input (sustained loops: the power cap applies)
python tools/debug/wino_layer_bench.py 2 direct,wino2,wino2h,wino2s,wino2p,wino4,wino6,wino2b 2>&1 | f > $O/${TAG}_wino_layers.txt
# 2. timing-only ablations of the shipped F(2,3) kernel and of F(6,3) (results wrong by construction)
{ cat build/ab/variants.txt
  for i in 0 1 2 3 4 5 6 7; do
    echo "--- ab$i: $(sed -n "$((i+1))p" build/ab/variants.txt)"
    BSVD_HIP_LIB=$R/build/ab/lib_ab$i.so WINO_LAYERS=0,1 python tools/debug/wino_layer_bench.py 1 wino2,wino6 2>&1 | f | grep -v BSVD_HIP_LIB | cut -c1-110
  done; } > $O/${TAG}_wino_ablation.txt
# 3. per-section cycle split (timeline build)
{ for a in "wino2 256 256 135 240 10" "wino2 128 128 270 480 10" "wino6 256 256 135 240 10" "wino2h 256 256 135 240 1" "wino2p 256 256 135 240 10"; do
    BSVD_HIP_LIB=$R/build/ab/lib_ab8.so python tools/debug/wx_timeline.py $a 2>&1 | f | grep -v BSVD_HIP_LIB
  done; } > $O/${TAG}_wino_timeline.txt
# 4. MFMA / VALU co-issue microbenchmark
[ -x build/mfma_valu ] && ./build/mfma_valu > $O/${TAG}_ubench_mfma_valu.txt 2>&1
# 5. PMC passes over the two 10-frame temporal-fusion layers: direct vs F(2,3) vs F(6,3)
WINO_LAYERS=0,1 bash tools/debug/wino_pmc.sh ${TAG}wino direct,wino2,wino6 > /dev/null 2>&1
cp $O/pmc_${TAG}wino/summary.txt $O/${TAG}_wino_pmc_summary.txt
# 6. the whole C1 clip per form, interleaved
{ for round in 1 2; do for form in direct wino2 wino4 wino6; do
    echo -n "[$round] $form: "
    python bench.py --wide-conv $form --no-cpu-baseline --no-power-probe --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%.1f frames/s  parity vs exact fp32 %.2e  dominant %s %.3f ms avg, %.0f TFLOP/s algorithmic, %.0f issued' % (d['value'], d['parity']['max_abs_f16x3_vs_exact_fp32_on_this_clip'], r['kernel'], r['avg_launch_ms'], r['achieved'], r['mfma_pipe_frac'] * r['peak']))"
  done; done; } > $O/${TAG}_c1_forms.txt
tail -n 100 $O/${TAG}_wino_layers.txt $O/${TAG}_wino_ablation.txt $O/${TAG}_wino_timeline.txt $O/${TAG}_c1_forms.txt
