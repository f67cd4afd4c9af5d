#!/bin/bash
cd $GRAFT_REPO_ROOT
for t in _c119 _c23 _c7; do echo "== const ABL $t"; CSD_LIB_PATH=$GRAFT_REPO_ROOT/conditional_score_diffusion_amd/libcsd_hip_tune$t.so ONLY=1 REPS=10 python tools/ff_probe.py 2>&1 | grep -v amdgpu; done > gpurun_out/iter2.txt
