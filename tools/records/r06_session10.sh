#!/bin/bash
# Round 6, GPU session 10: the C2 record re-taken (the final-record session's C2 line had a slow timed region: 444.8 frames/s burst against 492.8 sustained in the same process)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; export PYTHONDONTWRITEBYTECODE=1
for i in 1 2 3; do python bench.py --workload c2 --no-cpu-baseline 2>/dev/null > $O/r06_c2_try$i.json; python -c "
import json; d=json.load(open('$O/r06_c2_try$i.json')); print('try $i: burst %.1f sustained %.1f' % (d['value'], d['sustained']['value']), {k.replace('conv3x3_kernel','').replace('winox_kernel',''):round(v['ms_per_step'],1) for k,v in d['roofline']['all_conv_kernels'].items()})"; done
python bench.py --workload c2 > $O/r06_c2.json 2> $O/r06_c2.err; python -c "
import json; d=json.load(open('$O/r06_c2.json')); print('record: burst %.1f sustained %.1f cpu %.2f' % (d['value'], d['sustained']['value'], d['cpu_baseline']['value']))"
