cd $GRAFT_REPO_ROOT
./build/xp_order_probe | head -12
for sh in 2 1; do CSD_LIB_PATH=$GRAFT_REPO_ROOT/conditional_score_diffusion_amd/libcsd_hip_tune.so timeout 300 python tools/xp_timing.py $sh 2>&1 | grep -v amdgpu.ids; done
