cd $GRAFT_REPO_ROOT
export CSD_LIB_PATH=$GRAFT_REPO_ROOT/conditional_score_diffusion_amd/libcsd_hip_tune.so
for a in ${ABLS:-0 1 2 4 8 16 32 64 96 98 114 115}; do echo "== ABL $a"; CSD_FF_ABL=$a ONLY=${ONLY:-1} REPS=10 PREC=${PREC:-fp16f8} python tools/ff_probe.py 2>&1 | grep -v amdgpu; done > gpurun_out/abl.txt
