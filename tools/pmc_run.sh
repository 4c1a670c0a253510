#!/bin/bash
# Collects PMC counters for the bench workload in separate passes (guide: FETCH_SIZE and WRITE_SIZE cannot share
# a pass; counters never combined with sys/hip tracing).  Usage on the GPU box: tools/pmc_run.sh <tag>
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pass$i -o pmc -- \
     python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-power-probe --no-box-calibration ${BENCH_ARGS} > $OUT/pass$i.log 2>&1
  echo "pass $i ($set): rc=$?"
done
find $OUT -name "*.csv" | head -20
