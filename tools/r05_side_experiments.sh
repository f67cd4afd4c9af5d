#!/bin/bash
# round-5 side experiments: (1) NCSN++-256 forward at B = 8 / 16 / 32 ("benchmark it filled"); (2) the NIN shortcuts on
# a side stream behind Conv_0, re-measured with conv_xw (item 4): tuning build, CSD_SIDE_STREAM = 0 | 1, same box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python - <<'PY' 2>&1 | grep -v amdgpu | tee $O/r05_ncsnpp256_batch.txt
import sys; sys.argv = ['x', 'fp16x3']
sys.path.insert(0, 'tools')
import bench_other as b
for B in (8, 16, 32):
    try:
        b.ncsnpp256(B)
    except Exception as e:
        print('B = %d: %s' % (B, str(e)[:300]))
PY
for s in 0 1 0 1; do
  echo "CSD_SIDE_STREAM=$s"
  CSD_LIB_PATH=$R/conditional_score_diffusion_amd/libcsd_hip_tune.so CSD_SIDE_STREAM=$s python bench.py --steps 20 --warmup 3 --no-alt --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   %.4f img/s  %.2f ms per PC step' % (d['value'], d['ms_per_step']))"
done 2>&1 | tee $O/r05_side_stream.txt
