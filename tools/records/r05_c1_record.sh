R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/r05_c1.json 2> $OUT/r05_c1.err; tail -c 600 $OUT/r05_c1.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r05_c1 -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-power-probe > /dev/null 2> $OUT/prof_r05_c1.log
f=$(find $OUT/prof_r05_c1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r05_c1_kernel_stats.csv; rm -rf $OUT/prof_r05_c1
grep winox $OUT/r05_c1_kernel_stats.csv | head -2
