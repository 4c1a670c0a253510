#!/bin/bash
# The round's final records in ONE box session: the whole GPU suite + every profiles/r05_* record (tools/round_profiles.sh), then the C1 line
# re-taken against the traffic table of this very session (bench.py reads profiles/traffic.json), then the single-frame layer table.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
bash tools/round_profiles.sh r05
cp $O/r05_traffic.json $R/profiles/traffic.json
bash tools/r05_c1_record.sh
cd $R; python tools/per_layer_stream.py 2>&1 | grep -v "amdgpu.ids" > $O/r05i_per_layer_stream.txt; tail -1 $O/r05i_per_layer_stream.txt
