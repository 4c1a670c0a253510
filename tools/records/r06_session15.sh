#!/bin/bash
# Round 6, GPU session 15: L1 (TCP) / L2 (TCC) counters of the C1 clip's kernels: is the per-CU L2-fetch throughput what the Winograd kernel sits on? (the stride-2 tile: 41-51 % TCP stall)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/pmc_r06l1; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pass$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-power-probe --no-box-calibration > $O/pass$i.log 2>&1
  echo "pass $i rc=$?"
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $O > $GRAFT_REPO_ROOT/gpurun_out/r06_pmc_l1_summary.txt 2>&1
grep -v "^   SQ_\|=>" $GRAFT_REPO_ROOT/gpurun_out/r06_pmc_l1_summary.txt | head -120
