#!/bin/bash
# what the driver runs at round end, on the final tree: the GPU suite, smoke(), the default bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_verify; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -6
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; python -c "
import json; d=json.load(open('$O/bench.json')); print(round(d['value'],1), round(d['value_normalised'],1), round(d['roofline']['frac'],4), d['box_calibration']['direct64']['ms_per_launch'], d['roofline']['traffic_ratio'], d['cpu_baseline']['value'])"
