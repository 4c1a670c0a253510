#!/bin/bash
# L1 (TCP) / L2 (TCC) request counters of the bench kernels in two extra PMC passes: how many of a kernel's vector-memory requests
# miss the L1 and go to the L2, and how many of those miss the L2.  usage (GPU box): tools/pmc_l1.sh <tag>; summary: tools/pmc_summary.py
TAG=${1:-l1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pass$i -o pmc -- \
     python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-power-probe --no-box-calibration ${BENCH_ARGS} > $OUT/pass$i.log 2>&1
  echo "pass $i ($set): rc=$?"
done
