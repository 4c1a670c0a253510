#!/bin/bash
# Round-5 GPU session 14: shader clock and package power while ONE 256 -> 256 layer loops: shipped Winograd kernel, its ablation builds (no transform /
# no MFMA steps / neither) and the direct tile.   needs the variant libraries of tools/r05_session13.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
f() { grep -v "amdgpu.ids\|BSVD_HIP_LIB"; }
{ for i in 0 1 2 3; do echo -n "ab$i ($(sed -n "$((i+1))p" build/ab/variants.txt)): "; BSVD_HIP_LIB=$R/build/ab/lib_ab$i.so timeout 300 python tools/debug/layer_power.py wino2 256 135 240 10 6 2>&1 | f | tail -1; done
  echo -n "direct tile: "; timeout 300 python tools/debug/layer_power.py direct 256 135 240 10 6 2>&1 | f | tail -1
  for i in 0 1; do echo -n "128->128 ab$i: "; BSVD_HIP_LIB=$R/build/ab/lib_ab$i.so timeout 300 python tools/debug/layer_power.py wino2 128 270 480 10 6 2>&1 | f | tail -1; done; } > $O/r05_layer_power.txt 2>&1
cat $O/r05_layer_power.txt
