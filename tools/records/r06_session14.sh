#!/bin/bash
# Round 6, GPU session 14: timing-only ablations of the F(6,3) kernel on a transformed-domain input (its K loop is copy + MFMA steps): 1 no copy, 2 no MFMA steps (and no weight
# loads), 5 no copy + no epilogue finish, 8 no chunk barrier, 13 = 1 + 4 + 8
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r06_s14; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
for i in 0 1 2 3 4 5; do echo "== ab$i: $(sed -n "$((i+1))p" build/ab/variants.txt)"; BSVD_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/lib_ab$i.so V_LAYERS=0,1 timeout 300 python tools/debug/v_layer_bench.py 1.0 wino6,wino2 2>/dev/null | sed 's/| V in + V out.*//'; done | tee $O/v_consumer_ablation.txt
