#!/bin/bash
# Round profile on the GPU box: kernel-trace stats (CSV) of the bench command + PMC passes.  usage: profile_round.sh <tag>
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- \
   python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
echo "stats rc=$?"; find $OUT/stats -name "*kernel_stats.csv" | head -2
cd $GRAFT_REPO_ROOT && tools/pmc_run.sh $TAG
python tools/pmc_summary.py gpurun_out/pmc_$TAG > gpurun_out/pmc_$TAG/summary.txt
