#!/bin/bash
# Round 6, GPU session 7: stride-2 tile shape -- 64 px x 64 ch per wave (shipped) vs 128 px x 32 ch per wave (half the weight bytes per MFMA through the L1:
# TCP_PENDING_STALL_CYCLES says the L1 is stalled 41-51 % of the stride-2 kernels' time), padded and quad-planar patch.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_s7; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
for i in 0 1 2; do
  BSVD_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/lib_ab$i.so timeout 300 python -m pytest tests/test_gpu_f16x3.py -x -q -k "layer_split" 2>&1 | tail -1
  for L in "64 128 540 960 10 1.5 2" "128 256 270 480 10 1.5 2" "64 128 540 960 1 1 2" "128 256 270 480 1 1 2"; do
  echo -n "ab$i: "; BSVD_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/lib_ab$i.so python tools/debug/layer_loop.py $L 2>/dev/null; done; done | tee $O/stride2_tiles.txt
