#!/bin/bash
# Round-5 GPU session 1 (run on the GPU box through gpurun): the GPU suite on the round's first batch of changes, the driver's bench
# line, an interleaved A/B of the knobs those changes introduced, the bound of the fused 64-channel pair from timing-only ablation
# builds, the stride-2 tile's conflict-free layout with PMC counters, the c32 fold-8 occupancy.
# needs: tools/build_ab.sh "" "-DBSVD_FP16_OVFL=0" "-DBSVD_WX_MIXASM=2" "-DBSVD_TUNE_APFL=3" "-DBSVD_TUNE_QPL=7" "-DBSVD_ABL=32" "-DBSVD_ABL=2" "-DBSVD_TUNE_FOLD8_OCC=2" "-DBSVD_EPI_CLAMP=0"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
f() { grep -v amdgpu.ids; }
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | f | tail -40 > $O/r05a_gpu_tests.txt
tail -5 $O/r05a_gpu_tests.txt
python bench.py 2> $O/r05a_c1.err | tail -1 > $O/r05a_c1.json
python - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05a_c1.json")); r=d["roofline"]
print("C1 %.1f frames/s sustained %.1f frac %.4f traffic_alg %.3g ratio %s power %s" % (d["value"], d.get("sustained",{}).get("value",0), r["frac"], r["traffic_algorithmic"], r["traffic_ratio"], d.get("power")))
print({k: round(v["ms_per_step"],3) for k,v in r["all_conv_kernels"].items()})
PY
# interleaved A/B: ab0 product | ab1 no FP16_OVFL | ab2 re-split asm without its wait state | ab3 narrow tile on the same-register prefetch | ab4 stride-2 quad-planar | ab8 no explicit clamp in the split stores
{ cat build/ab/variants.txt
  for round in 1 2; do for i in 0 1 2 3 4 8; do
    echo -n "[$round] ab$i: "
    BSVD_HIP_LIB=$R/build/ab/lib_ab$i.so python bench.py --no-cpu-baseline --no-power-probe --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('fps %.1f parity %.2e' % (d['value'], d['parity']['max_abs_f16x3_vs_exact_fp32_on_this_clip']), {k.replace('conv3x3_kernel','').replace('winox_kernel','wx'):round(v['ms_per_step'],3) for k,v in r['all_conv_kernels'].items()})"
  done; done; } > $O/r05a_ab.txt 2>&1
cat $O/r05a_ab.txt
# bound of a fused 64 -> 64 pair from the timing-only ablation builds: first conv without its output stores, second conv
# without its activation loads, each on realistic operands, sustained loops
{ echo "# 64->64 at 540x960 x10, ms per launch: shipped | no epilogue stores (ABL 32) | no activation loads in the K loop (ABL 2)"
  for i in 0 5 6 0 5 6; do BSVD_HIP_LIB=$R/build/ab/lib_ab$i.so python tools/debug/layer_loop.py 64 64 540 960 10 3 2>&1 | f | grep -v BSVD_HIP_LIB; done
  echo "# 64->3 exit-shaped layer is not loopable here (planar out); 128->128 270x480 x10 TSM for scale:"
  for i in 0 5 6; do BSVD_HIP_LIB=$R/build/ab/lib_ab$i.so python tools/debug/layer_loop.py 128 128 270 480 10 2 1 1 2>&1 | f | grep -v BSVD_HIP_LIB; done
} > $O/r05a_fused_pair_bound.txt 2>&1
cat $O/r05a_fused_pair_bound.txt
# stride-2 tile: padded (shipped) vs quad-planar layout, LDS conflict counters
cd /tmp && export TMPDIR=/tmp
for v in 0 4; do
  for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
    n=pass$(echo $set | wc -w)
    BSVD_HIP_LIB=$R/build/ab/lib_ab$v.so rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_r05a_s2_ab$v/$n -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-power-probe > $O/pmc_r05a_s2_ab$v.$n.log 2>&1
  done
done
cd $R
for v in 0 4; do echo "== ab$v"; python tools/pmc_summary.py $O/pmc_r05a_s2_ab$v 2>/dev/null | grep -A16 "ConvCfg<2, 2, 2, 2, 2" | head -20; done > $O/r05a_stride2_qpl_pmc.txt 2>&1
cat $O/r05a_stride2_qpl_pmc.txt | tail -45
# c32-sized network: fold-8 instantiation at 3 (20-byte preheader spill) vs 2 waves per SIMD
for i in 0 7 0 7; do echo -n "ab$i: "; BSVD_HIP_LIB=$R/build/ab/lib_ab$i.so python tools/c32_fps.py 2>&1 | f | grep "f16x3 clip"; done > $O/r05a_c32_fold8.txt 2>&1
cat $O/r05a_c32_fold8.txt
