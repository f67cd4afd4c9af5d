#!/bin/bash
# conv_xp bring-up on the GPU box: op-level parity, per-layer timings A/B against conv_ff, ablations, in-kernel stamps
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/xp; mkdir -p $O
cd $R
if [ -x build/xp_order_probe ]; then ./build/xp_order_probe | tee $O/order_probe.txt; fi
timeout 300 python tools/xp_debug.py 2>&1 | grep -v amdgpu.ids > $O/debug.txt; grep -v "per \|stats" $O/debug.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "conv3x3_block" > $O/pytest_block.txt 2>&1
tail -3 $O/pytest_block.txt
REPS=20 PREC=fp16x3 timeout 300 python tools/ff_probe.py 2>&1 | grep -v amdgpu.ids > $O/probe_xp.txt
echo "--- xp"; cat $O/probe_xp.txt
for a in "$@"; do
  echo "--- abl $a"
  CSD_LIB_PATH=$R/conditional_score_diffusion_amd/libcsd_hip_abl$a.so ONLY=1,2 REPS=20 PREC=fp16x3 timeout 300 python tools/ff_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/probe_abl$a.txt
done
for sh in 1 2; do
CSD_LIB_PATH=$R/conditional_score_diffusion_amd/libcsd_hip_tune.so timeout 300 python tools/xp_timing.py $sh 2>&1 | grep -v amdgpu.ids | tee $O/timing_$sh.txt
done
