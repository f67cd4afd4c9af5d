#!/bin/bash
# tuning build (CSD_LIB_PATH = libcsd_hip_tune.so): the headline loop with and without the batch chunks of the <= 20^2 levels, by batch
set -u
R=$GRAFT_REPO_ROOT; cd $R
export CSD_LIB_PATH=$R/conditional_score_diffusion_amd/libcsd_hip_tune.so
for r in 1 2 3; do for b in ${BATCHES:-64}; do for k in ${KS:-1 2}; do
  v=$(CSD_CHUNKS=$k python bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline --no-profile --batch $b 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f'%j['ms_per_step'])")
  echo "round $r B=$b K=$k: $v ms/step"
done; done; done
