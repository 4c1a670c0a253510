#!/bin/bash
# Round-5 GPU session 2: the fused 64-channel pair -- parity first, then A/B against the two launches, PMC counters and
# HBM traffic of the fused kernels; the whole GPU suite (no -x: every failure is listed).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
f() { grep -v amdgpu.ids; }
timeout 900 python -m pytest tests/test_gpu_pair.py tests/test_gpu_native.py -q 2>&1 | f | tail -40 > $O/r05c_pair_range_tests.txt
tail -25 $O/r05c_pair_range_tests.txt
{ for round in 1 2 3; do for fp in off on; do
    echo -n "[$round] fuse_pairs=$fp: "
    python bench.py --fuse-pairs $fp --no-cpu-baseline --no-power-probe --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('fps %.1f parity %.2e' % (d['value'], d['parity']['max_abs_f16x3_vs_exact_fp32_on_this_clip']), {k.replace('conv3x3_kernel','').replace('winox_kernel','wx'):(round(v['ms_per_step'],3), v['launches_per_step']) for k,v in r['all_conv_kernels'].items()})"
  done; done; } > $O/r05c_fuse_pairs_ab.txt 2>&1
cat $O/r05c_fuse_pairs_ab.txt
# counters of the fused kernels (separate passes; FETCH_SIZE / WRITE_SIZE each on their own)
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_r05c_pairs/pass$i -o pmc -- python $R/bench.py --fuse-pairs on --steps 1 --warmup 1 --no-cpu-baseline --no-power-probe > $O/pmc_r05c_pairs.pass$i.log 2>&1
done
cd $R
python tools/pmc_summary.py $O/pmc_r05c_pairs 2>/dev/null | grep -B1 -A18 "ConvCfg<4, 1, 2, 2, 1, 3, true>, true, 1, false, false\|ConvCfg<2, 1, 4, 1, 1" > $O/r05c_pairs_pmc_summary.txt 2>&1
cat $O/r05c_pairs_pmc_summary.txt | head -90
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | f | tail -30 > $O/r05c_gpu_tests.txt
tail -12 $O/r05c_gpu_tests.txt
