#!/bin/bash
# full GPU test suite + the fp16x3 bench line + one-evaluation timeline
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04b; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
python bench.py --precision fp16x3 --no-alt --no-cpu-baseline > $O/bench_fp16x3.json 2> $O/bench_fp16x3.err
python - <<'PY'
import json
j = json.loads(open('gpurun_out/r04b/bench_fp16x3.json').read().strip().splitlines()[-1])
print('value', j['value'], 'ms/step', j['ms_per_step'], j['kernel_classes_ms_per_step'])
PY
bash tools/timeline_run.sh fp16x3 r04b_fp16x3
cp gpurun_out/timeline_r04b_fp16x3.txt $O/
grep -E "conv_xp|conv_ff|kernel time" gpurun_out/timeline_r04b_fp16x3.txt | cut -c1-110
