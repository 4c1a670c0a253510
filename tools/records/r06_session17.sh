#!/bin/bash
# Round 6, GPU session 17: the persistent form (next tile's prologue under the last MFMA steps of the current one) on the round's final sources -- an input of profiles/r06_wino128_bound.txt
cd $GRAFT_REPO_ROOT; export PYTHONDONTWRITEBYTECODE=1
BSVD_HIP_LIB=$GRAFT_REPO_ROOT/build/measure/libbsvd_hip.so WINO_LAYERS=0,1 python tools/debug/wino_layer_bench.py 1.5 wino2,wino2p,wino2s 2>/dev/null | sed 's/max-abs vs wino2/|/' | tee gpurun_out/r06_persistent_recheck.txt
