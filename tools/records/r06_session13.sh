#!/bin/bash
# Round 6, GPU session 13: tile order of the Winograd kernel -- channel tile fastest (shipped: the channel tiles of a pixel tile are concurrent neighbours, its input is read once) vs
# channel tile slowest within a frame (an XCD works on ONE channel tile's U at a time: the U of the 256 -> 512 layer, 6.3 MB for F(2,3), does not fit a 4 MB L2)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r06_s13; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
for r in 1 2; do for i in 0 1; do echo "== [$r] ab$i"; BSVD_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/lib_ab$i.so WINO_LAYERS=0,1,2,3 timeout 300 python tools/debug/wino_layer_bench.py 1.0 wino2,wino6 2>/dev/null | sed 's/max-abs.*//'; done; done | tee $O/ctouter_layers.txt
timeout 600 tools/ab_prebuilt.sh 2 2 2>&1 | cut -c1-260 | tee $O/ctouter_c1.txt
