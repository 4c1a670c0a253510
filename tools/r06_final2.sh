#!/bin/bash
# Round 6, final records, part 2: the profiler passes again WITHOUT the box-calibration loops in the profiled process (part 1's PMC / kernel-stats passes
# averaged ~1000 launches of the small calibration layer into the Winograd kernel's per-launch figures), then the C1 line against this session's traffic table.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
export PYTHONDONTWRITEBYTECODE=1
bash tools/bench_workloads.sh r06 2>&1 | grep "rc="
bash tools/pmc_run.sh r06 > $O/pmc_r06.log 2>&1
python tools/pmc_summary.py $O/pmc_r06 > $O/r06_pmc_summary.txt
python tools/make_traffic.py $O/pmc_r06 $O/r06_traffic.json > /dev/null
cp $O/r06_traffic.json $R/profiles/traffic.json
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/r06_c1.json 2> $O/r06_c1.err; tail -c 300 $O/r06_c1.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r06_c1 -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-power-probe --no-box-calibration > /dev/null 2> $O/prof_r06_c1.log
f=$(find $O/prof_r06_c1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r06_c1_kernel_stats.csv; rm -rf $O/prof_r06_c1
grep -i "winox" $O/r06_c1_kernel_stats.csv | head -3
