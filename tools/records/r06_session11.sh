#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONDONTWRITEBYTECODE=1
./tests/native/abi_parity.bin 2>&1 | tail -14
timeout 600 python -m pytest tests/test_gpu_native.py tests/test_gpu_dist.py -x -q 2>&1 | tail -3
