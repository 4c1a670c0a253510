#!/bin/bash
# Round-5 GPU session 7: the whole GPU suite with the plain-fp32 hand-over on by default; stream schedules with it on / off on one box.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
f() { grep -v amdgpu.ids; }
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | f | tail -30 > $O/r05f_gpu_tests.txt
tail -8 $O/r05f_gpu_tests.txt
for h in off on off on; do echo "== f32_handover=$h"; ONLY="clip,graphs (per-frame,chunk 1: lagged,chunk 8: lagged" python tools/stream_modes.py --frames 85 --f32-handover $h 2>&1 | f | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('%-55s %.1f %s' % (r['schedule'], r['fps'], r['bitwise_equal_to_clip']))"; done > $O/r05f_stream_modes_handover.txt 2>&1
cat $O/r05f_stream_modes_handover.txt
