#!/bin/bash
# Everything profiles/<tag>_* is made of, in one GPU-box session.  usage: tools/round_profiles.sh <tag>
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/${TAG}_gpu_tests.txt; cat $O/${TAG}_gpu_tests.txt
bash tools/bench_workloads.sh $TAG 2>&1 | grep "rc="
bash tools/pmc_run.sh $TAG > $O/pmc_$TAG.log 2>&1
python tools/pmc_summary.py $O/pmc_$TAG > $O/${TAG}_pmc_summary.txt
python tools/make_traffic.py $O/pmc_$TAG $O/${TAG}_traffic.json > /dev/null
python tools/stream_modes.py --frames 85 --json $O/${TAG}_stream_modes_540x960.json 2>&1 | tail -14
python tools/c5_stream.py --json $O/${TAG}_c5_stream_f16x3.json 2>&1 | tail -3
python tools/full_clip_parity.py --workloads c2,c3 --json $O/${TAG}_full_clip_parity.json 2>&1 | tail -2 | cut -c1-400
ls $O | grep "^${TAG}_"
