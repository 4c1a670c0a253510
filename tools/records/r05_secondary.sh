#!/bin/bash
# Round-5 secondary records on the final code (rows of README.md that earlier rounds measured): live host-to-host feeds, the PCIe-inclusive
# clip pipeline, C3 as the reference runs it (TSN + denoise_seq), the c32-sized network, the 4K stream, a camera wall; + the new folded-band fuzz.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
f() { grep -v "amdgpu.ids\|BSVD_HIP_LIB"; }
timeout 600 python -m pytest tests/test_gpu_wino.py -q -x 2>&1 | f | tail -3
{ echo "== live_stream 540x960";  timeout 600 python tools/live_stream.py --size 540x960 --frames 200 2>&1 | f | tail -4
  echo "== live_stream 1080x1920"; timeout 600 python tools/live_stream.py 2>&1 | f | tail -4
  echo "== pcie_pipeline"; timeout 600 python tools/pcie_pipeline.py 2>&1 | f | tail -4
  echo "== mimo_fps (C3 through TSN + denoise_seq)"; timeout 900 python tools/mimo_fps.py 2>&1 | f | tail -4
  echo "== c32_fps"; timeout 600 python tools/c32_fps.py 2>&1 | f | tail -4
  echo "== concurrent_streams"; timeout 900 python tools/concurrent_streams.py 2>&1 | f | tail -6
  echo "== stream_4k"; timeout 1200 python tools/stream_4k.py 2>&1 | f | tail -5
  echo "== c5_stream"; timeout 900 python tools/c5_stream.py --json $O/r05_c5_stream_f16x3.json 2>&1 | f | tail -3; } > $O/r05_secondary.txt 2>&1
cat $O/r05_secondary.txt
