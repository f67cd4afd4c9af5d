#!/bin/bash
cd $GRAFT_REPO_ROOT
export CSD_LIB_PATH=$GRAFT_REPO_ROOT/conditional_score_diffusion_amd/libcsd_hip_tune.so
export CSD_FF_LDS_PAD=20000
(ONLY=1 REPS=10 python tools/ff_probe.py 2>&1 | grep -v amdgpu) > gpurun_out/iter3.txt
cd tools; (python ff_timing.py fp16x3 1; python ff_timing.py fp16 1) 2>&1 | grep -v amdgpu >> ../gpurun_out/iter3.txt
