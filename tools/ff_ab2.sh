#!/bin/bash
cd $GRAFT_REPO_ROOT
(timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k conv3x3_block 2>&1 | tail -3) > gpurun_out/iter_test.log
for rep in 1 2; do
echo "== persistent (rep $rep)"; ONLY=${ONLY:-1,2,3} REPS=10 python tools/ff_probe.py 2>&1 | grep -v amdgpu
echo "== one workgroup per item (rep $rep)"; CSD_FF_NOPERSIST=1 ONLY=${ONLY:-1,2,3} REPS=10 python tools/ff_probe.py 2>&1 | grep -v amdgpu
done > gpurun_out/ab.txt
