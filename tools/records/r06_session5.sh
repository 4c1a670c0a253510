#!/bin/bash
# Round 6, GPU session 5: what the stride-2 tile waits on: per-workgroup timeline of both stride-2 layers (-DBSVD_TIMELINE build)
# and the wave-time / instruction / memory counters over a loop of each layer; + the transformed-domain consumer on its final unit order.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_s5; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
{ for a in "64 128 540 960 10 2" "128 256 270 480 10 2" "64 128 540 960 1 2" "128 256 270 480 1 2"; do
    echo "== timeline $a"; BSVD_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/lib_ab0.so timeout 200 python tools/timeline.py $a 2>/dev/null
  done; } > $O/stride2_timeline.txt 2>&1; cat $O/stride2_timeline.txt
V_LAYERS=0,1 timeout 300 python tools/debug/v_layer_bench.py 1.5 wino6 2>/dev/null | sed 's/max-abs.*//' | tee $O/v_layers_final.txt
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  for L in "64 128 540 960 10 0.3 2" "128 256 270 480 10 0.3 2"; do
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc/pass$i -o pmc_$(echo $L | tr ' ' _) -- python $GRAFT_REPO_ROOT/tools/debug/layer_loop.py $L > $O/pmc_pass$i.log 2>&1
  done
  echo "pass $i rc=$?"
done
python - <<'PY'
import csv, glob, collections, os
O=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/r06_s5"
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(O+"/pmc/pass*/**/*counter_collection.csv", recursive=True):
    layer="64->128" if "64_128" in f else "128->256"
    for r in csv.DictReader(open(f)):
        if "ConvCfg<2, 2, 2, 2, 2" not in r["Kernel_Name"]: continue
        agg[layer][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[layer][r["Counter_Name"]]+=1
with open(O+"/stride2_pmc.txt","w") as out:
    for layer,v in agg.items():
        print("stride 2,", layer, file=out)
        for c,x in sorted(v.items()): print("   %-34s per dispatch %.4g" % (c, x/cnt[layer][c]), file=out)
print(open(O+"/stride2_pmc.txt").read())
PY
