#!/bin/bash
# tuning iteration: unit test of the fused block convolution, probe timings (product library), phase stamps (tuning library)
cd $GRAFT_REPO_ROOT
(timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k conv3x3_block 2>&1 | tail -3) > gpurun_out/iter_test.log
(ONLY=${ONLY:-1,3} REPS=10 python tools/ff_probe.py 2>&1 | grep -v amdgpu) > gpurun_out/iter_probe.txt
export CSD_LIB_PATH=$GRAFT_REPO_ROOT/conditional_score_diffusion_amd/libcsd_hip_tune.so
cd tools; (python ff_timing.py fp16x3 1; python ff_timing.py fp16 1) 2>&1 | grep -v amdgpu > ../gpurun_out/iter_timing.txt
