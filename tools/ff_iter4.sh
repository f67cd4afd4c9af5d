#!/bin/bash
cd $GRAFT_REPO_ROOT
(timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k conv3x3_block 2>&1 | tail -4) > gpurun_out/iter_test.log
(echo "== persistent"; CSD_FF_PERSISTENT=1 ONLY=${ONLY:-0,1,2,3,4,5} REPS=10 timeout 120 python tools/ff_probe.py 2>&1 | grep -v amdgpu; echo "== default"; ONLY=${ONLY:-0,1,2,3,4,5} REPS=10 python tools/ff_probe.py 2>&1 | grep -v amdgpu) > gpurun_out/iter_probe.txt
