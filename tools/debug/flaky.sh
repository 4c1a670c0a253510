#!/bin/bash
# runs the lagged-graph stream==clip test N times per library variant
cd $GRAFT_REPO_ROOT
N=${1:-5}
for lib in ${LIBS:-lib_r02 lib_ab0 lib_oldhead}; do
  ok=0; bad=0
  for i in $(seq 1 $N); do
    if BSVD_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/$lib.so python -m pytest tests/test_gpu_stream_graph.py -m gpu -q -x -k "ring_graph_stream_equals_clip_bitwise and f16x3" > /tmp/flaky.log 2>&1; then ok=$((ok+1)); else bad=$((bad+1)); grep -E "^E  |assert" /tmp/flaky.log | head -3; fi
  done
  echo "$lib: pass $ok fail $bad"
done
