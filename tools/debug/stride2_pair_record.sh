#!/bin/bash
# stride-2 tile: line-pair staging variants (whole 128-B lines per fetch) vs the shipped register double buffer: time and HBM traffic
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
{
for i in 0 1 2; do
  echo "=== ab$i: $(sed -n "$((i+1))p" $R/build/ab/variants.txt)"
  for r in 1 2; do
  BSVD_HIP_LIB=$R/build/ab/lib_ab$i.so python $R/bench.py --no-cpu-baseline --no-power-probe --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['all_conv_kernels']['conv3x3_kernel<2,2,2,2,2>[f16x3]']
print('  C1 %.1f frames/s; stride-2 tile: %.3f ms per clip (4 launches), %.0f TFLOP/s' % (d['value'], k['ms_per_step'], k['tflops']))"
  done
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_s2_$i_$c
    BSVD_HIP_LIB=$R/build/ab/lib_ab$i.so rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_s2/ab$i/pass_$c -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-power-probe > /dev/null 2>&1
  done
  python - <<PY
import csv,glob,collections
agg=collections.defaultdict(float); n=collections.defaultdict(set)
for f in glob.glob("$O/pmc_s2/ab$i/pass_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "ConvCfg<2, 2, 2, 2, 2" in r["Kernel_Name"] and ", 1," in r["Kernel_Name"]:
            agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
fe=agg["FETCH_SIZE"]/max(len(n["FETCH_SIZE"]),1)*1024; wr=agg["WRITE_SIZE"]/max(len(n["WRITE_SIZE"]),1)*1024
print("  HBM per launch (mean of both stride-2 layers, %d launches): read 2 x FETCH_SIZE = %.3f GB, write %.3f GB, total %.3f GB" % (len(n["FETCH_SIZE"]), 2*fe/1e9, wr/1e9, (2*fe+wr)/1e9))
PY
done
} > $O/r04_stride2_pair_variants.txt 2>&1
cat $O/r04_stride2_pair_variants.txt
