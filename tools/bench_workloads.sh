#!/bin/bash
# The four single-GPU BASELINE configurations through bench.py, each with its rocprofv3 kernel-trace stats.
# usage on the GPU box: tools/bench_workloads.sh <tag>      -> gpurun_out/<tag>_<workload>[_mode].json + *_kernel_stats.csv
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() {   # name, bench args...
  local name=$1; shift
  python $R/bench.py "$@" > $OUT/${TAG}_$name.json 2> $OUT/${TAG}_$name.err
  echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$OUT/${TAG}_$name.json')); print('%.1f frames/s, dominant %s %.0f TF (frac %.3f), parity %.1e' % (d['value'], d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['frac'], d['parity']['max_abs_f16x3_vs_exact_fp32_on_this_clip']))" 2>&1 | tail -1)"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_$name -o bench -- \
     python $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-power-probe --no-box-calibration > /dev/null 2> $OUT/prof_${TAG}_$name.log
  f=$(find $OUT/prof_${TAG}_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_${name}_kernel_stats.csv
  rm -rf $OUT/prof_${TAG}_$name
}
run c1 --workload c1
run c1_stream --workload c1 --mode stream --no-cpu-baseline
run c2 --workload c2
run c2_stream --workload c2 --mode stream --no-cpu-baseline
run c3 --workload c3
run c5 --workload c5
run c5_clip --workload c5 --mode clip --no-cpu-baseline
ls -la $OUT | grep ${TAG}_
