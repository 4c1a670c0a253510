#!/bin/bash
# Round-5 GPU session 13: timing-only ablations of the FINAL F(2,3) kernel (fp32 input and output, folded band) -- results wrong by construction.
# needs: tools/build_ab.sh "" "-DBSVD_WX_ABL=1" "-DBSVD_WX_ABL=2" "-DBSVD_WX_ABL=3" "-DBSVD_WX_ABL=4" "-DBSVD_WX_ABL=8" "-DBSVD_WX_ABL=64"
# (1 no transform; 2 no MFMA steps; 3 neither; 4 no epilogue finish; 8 no chunk barrier; 64 no activation requests in the K loop)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
f() { grep -v "amdgpu.ids\|BSVD_HIP_LIB"; }
{ cat build/ab/variants.txt
  for i in 0 1 2 3 4 5 6; do echo "--- ab$i: $(sed -n "$((i+1))p" build/ab/variants.txt)"
    BSVD_HIP_LIB=$R/build/ab/lib_ab$i.so timeout 600 python tools/debug/wino_f32_bench.py 2 wino2 2>&1 | f | grep "in f32   out f32\|in pairs out pairs" | cut -c1-120; done; } > $O/r05_wino_ablation.txt 2>&1
cat $O/r05_wino_ablation.txt
