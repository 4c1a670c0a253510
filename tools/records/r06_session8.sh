#!/bin/bash
# Round 6, GPU session 8: stride-2 tile -- item map x wave tile, layers in a loop and the C1 clip interleaved (digests via parity column)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_s8; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
for r in 1 2; do for i in 0 1 2; do for L in "64 128 540 960 10 1.2 2" "128 256 270 480 10 1.2 2"; do
  echo -n "[$r] ab$i: "; BSVD_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/lib_ab$i.so python tools/debug/layer_loop.py $L 2>/dev/null; done; done; done | tee $O/stride2_variants.txt
for i in 0 1 2; do echo -n "ab$i digest: "; BSVD_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/lib_ab$i.so python bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-power-probe --no-box-calibration --scaling strong --total-frames 10 --output-digest 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['output_digest']['sha256_16_per_10_frame_block'], round(d['value'],1))"; done | tee $O/digests.txt
timeout 900 tools/ab_prebuilt.sh 3 2 2>&1 | tee $O/ab_stride2_variants.txt
