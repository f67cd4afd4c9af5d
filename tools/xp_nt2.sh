#!/bin/bash
# conv_xp with 64-cout groups (nf = 128 networks): parity, layer timings against conv_ff, the side benches that use them
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/nt2; mkdir -p $O
cd $R
timeout 300 python tools/xp_debug.py 2>&1 | grep -v amdgpu.ids > $O/debug.txt; grep "max err" $O/debug.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "conv3x3_block" > $O/pytest_block.txt 2>&1; tail -3 $O/pytest_block.txt
T=$R/conditional_score_diffusion_amd/libcsd_hip_tune.so
for xp in 0 1; do
  echo "--- CSD_XP=$xp (tuning library)"
  CSD_XP=$xp CSD_LIB_PATH=$T ONLY=0,1,2,4,6,7 REPS=20 PREC=fp16x3 timeout 300 python tools/ff_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/probe_xp$xp.txt
done
echo "--- product library"
ONLY=0,1,2,4,6,7 REPS=20 PREC=fp16x3 timeout 300 python tools/ff_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/probe_prod.txt
timeout 600 python tools/bench_other.py 2>&1 | grep -v amdgpu.ids | tee $O/other.txt
timeout 600 python tools/bench_train.py --model ddpm_paired 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/train.txt
timeout 600 python bench.py --steps 20 --warmup 3 --cpu-steps 0 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/bench.txt
timeout 1500 python -m pytest tests/test_gpu_network.py tests/test_ncsnpp.py tests/test_gpu_training.py -x -q > $O/pytest_net.txt 2>&1; tail -3 $O/pytest_net.txt
