#!/bin/bash
cd $GRAFT_REPO_ROOT
export CSD_LIB_PATH=$GRAFT_REPO_ROOT/conditional_score_diffusion_amd/libcsd_hip_tune.so
cd tools; (CSD_FF_ABL=256 python ff_timing.py fp16x3 1; echo "--- single WG per CU"; CSD_FF_LDS_PAD=20000 CSD_FF_ABL=256 python ff_timing.py fp16x3 1) 2>&1 | grep -v amdgpu > ../gpurun_out/iter6.txt
