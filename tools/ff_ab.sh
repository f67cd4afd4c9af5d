#!/bin/bash
# in-run A/B of library variants on the fused block convolution probe (variants = conditional_score_diffusion_amd/libcsd_hip_<tag>.so)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for t in "" $TAGS; do
  lib=$GRAFT_REPO_ROOT/conditional_score_diffusion_amd/libcsd_hip${t:+_$t}.so
  echo "== ${t:-default} (rep $rep)"; CSD_LIB_PATH=$lib ONLY=${ONLY:-1,2,3} REPS=10 PREC=${PREC:-fp16x3} python tools/ff_probe.py 2>&1 | grep -v amdgpu
done; done > gpurun_out/ab.txt
