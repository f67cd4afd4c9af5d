#!/bin/bash
# round-6 quick loop (GPU box): the network-level parity tests + the headline at B = 64 (and smaller per-GPU batches with $1 = scan)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06q; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_network.py tests/test_gpu_steps.py tests/test_gpu_sharded.py -x -q -m gpu 2>&1 | tail -8 > $O/tests.txt
cat $O/tests.txt
for b in ${BATCHES:-64}; do
  python bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline --batch $b > $O/bench_b$b.json 2> $O/bench_b$b.err
  python - <<P
import json
try:
    j=json.loads(open('$O/bench_b$b.json').read().strip().splitlines()[-1])
    print('B=$b value %.4f img/s  %.2f ms/step'%(j['value'],j['ms_per_step']), {k:round(v,2) for k,v in j['kernel_classes_ms_per_step'].items()})
except Exception as e: print('$b failed',e); print(open('$O/bench_b$b.err').read()[-2000:])
P
done
