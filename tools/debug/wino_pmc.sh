#!/bin/bash
# PMC passes over tools/debug/wino_layer_bench.py (short loops): MFMA pipe busy, wave-time split, LDS conflicts per kernel.
# usage on the GPU box: tools/debug/wino_pmc.sh <tag> [forms] ; env WINO_LAYERS picks the layers
TAG=${1:-wino}
FORMS=${2:-direct,wino2,wino4}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pass$i -o pmc -- \
     python $GRAFT_REPO_ROOT/tools/debug/wino_layer_bench.py 0.15 $FORMS > $OUT/pass$i.log 2>&1
  echo "pass $i ($set): rc=$?"
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
