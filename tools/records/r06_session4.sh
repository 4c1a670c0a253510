#!/bin/bash
# Round 6, GPU session 4: where the transformed-domain epilogue's +0.07 .. +0.11 ms per launch goes: timing-only ablation builds of the producer
# (no plane stores / no edge-record stores / no patch pass / none of the three / no epilogue finish at all) on the two temporal layers, and
# a rocprofv3 kernel trace of the C1 clip with the hand-over on (the patch kernel's own time).
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_s4; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
for i in 0 1 2 3 4 5; do
  echo "== ab$i: $(sed -n "$((i+1))p" build/ab/variants.txt)"
  BSVD_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/lib_ab$i.so V_LAYERS=0,1 timeout 300 python tools/debug/v_layer_bench.py 1.0 wino6 2>/dev/null | sed 's/max-abs.*//'
done 2>&1 | tee $O/v_epilogue_ablation.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o c1_v -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 2 --no-power-probe --no-box-calibration --wide-conv wino6 --v-handover on > $O/bench_v_prof.json 2> $O/bench_v_prof.err
find $O/prof -name "*kernel_stats*" | head; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -16 "$f" | cut -c1-220
