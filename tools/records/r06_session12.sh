#!/bin/bash
# Round 6, GPU session 12: PMC passes + kernel trace over tools/debug/v_layer_bench.py (short loops): MFMA pipe busy, wave-time split, LDS conflicts, HBM bytes of the
# F(6,3) kernel with its transform in the K loop / with a transformed-domain input / with a transformed-domain output too, and the patch pass's own duration.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_r06v; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  V_LAYERS=0,1 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pass$i -o pmc -- python $GRAFT_REPO_ROOT/tools/debug/v_layer_bench.py 0.15 wino6 > $O/pass$i.log 2>&1
  echo "pass $i rc=$?"
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $O > $GRAFT_REPO_ROOT/gpurun_out/r06_v_pmc_summary.txt 2>&1
V_LAYERS=0,1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $GRAFT_REPO_ROOT/tools/debug/v_layer_bench.py 0.3 wino6 > $O/trace.log 2>&1
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/r06_v_kernel_stats.csv
grep -i "winox\|v_patch\|to_v" $GRAFT_REPO_ROOT/gpurun_out/r06_v_kernel_stats.csv | cut -c1-170
grep "^winox\|^v_patch\|=>\|FETCH\|WRITE" $GRAFT_REPO_ROOT/gpurun_out/r06_v_pmc_summary.txt | head -60
