#!/bin/bash
# Round 6, GPU session 6: the stride-2 tile's new item map (two patch rows per pass: one v_add per item at a chunk boundary instead of a full decode, twice)
# against the flattened map: parity tests of the stride-2 layers, the layers in a loop, the C1 clip interleaved.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_s6; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_f16x3.py tests/test_gpu_parity.py tests/test_gpu_f32_handover.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -4 | tee $O/pytest.txt
for i in 0 1; do for L in "64 128 540 960 10 1.5 2" "128 256 270 480 10 1.5 2" "64 128 540 960 1 1 2" "128 256 270 480 1 1 2"; do
  echo -n "ab$i: "; BSVD_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/lib_ab$i.so python tools/debug/layer_loop.py $L 2>/dev/null; done; done | tee $O/stride2_layers.txt
timeout 900 tools/ab_prebuilt.sh 2 2 2>&1 | tee $O/ab_stride2.txt
