#!/bin/bash
# On the GPU box: runs the FETCH_SIZE calibration (tools/traffic_cal/cal.hip; build it in the container first:
#   /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/traffic_cal/cal.hip -o tools/traffic_cal/cal.bin ) under
# rocprofv3 and writes gpurun_out/traffic_calibration.json.   usage: tools/traffic_cal/run.sh
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/traffic_cal
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 120 $R/tools/traffic_cal/cal.bin 3 > $OUT/plain.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc -o cal -- $R/tools/traffic_cal/cal.bin 3 > $OUT/cal.log 2> $OUT/cal.err
echo "rc=$?"; cat $OUT/cal.log
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc2 -o cal -- $R/tools/traffic_cal/cal.bin 3 > $OUT/cal2.log 2> $OUT/cal2.err
echo "rc2=$?"
python $R/tools/traffic_cal/parse.py $OUT $R/gpurun_out/traffic_calibration.json
