#!/bin/bash
# Weak-scaling table of the sampling bench on ONE node: bench.py at N = 1, 2, 4, 8 GPUs (one process per GPU over RCCL, 64 images per GPU),
# exactly as the driver launches it.  Usage: bash tools/scale.sh [steps] [warmup] [port]   ->  gpurun_out/scale/scale.json + a table
set -u
cd "$(dirname "$0")/.."
STEPS=${1:-20}; WARM=${2:-5}; PORT=${3:-29517}
O=gpurun_out/scale; mkdir -p $O
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
for N in 1 2 4 8; do
  if [ "$N" -gt "$NGPU" ]; then echo "N=$N: only $NGPU GPU(s) visible - skipped"; continue; fi
  if [ "$N" -eq 1 ]; then
    python bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-alt --no-cpu-baseline > $O/n$N.json 2> $O/n$N.err
  else
    HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
      bench.py --gpus $N --steps $STEPS --warmup $WARM --no-alt --no-cpu-baseline > $O/n$N.json 2> $O/n$N.err
  fi
done
python - <<'PY'
import json, glob, os
rows = []
for f in sorted(glob.glob('gpurun_out/scale/n*.json')):
    lines = [l for l in open(f).read().strip().splitlines() if l.startswith('{')]
    if lines:
        j = json.loads(lines[-1])
        rows.append((j['n_gpus'], j['value'], j['ms_per_step'], j['hbm_roofline']['frac']))
if rows:
    base = rows[0][1] / rows[0][0]
    print('%4s %12s %10s %10s %12s' % ('GPUs', 'images/s', 'ms/step', 'x N=1', 'HBM-roof frac'))
    for n, v, ms, fr in rows:
        print('%4d %12.3f %10.2f %10.2f %12.3f' % (n, v, ms, v / (base * 1), fr))
    json.dump([{'n_gpus': n, 'images_per_sec': v, 'ms_per_step': ms, 'hbm_roofline_frac_per_gpu': fr} for n, v, ms, fr in rows],
              open('gpurun_out/scale/scale.json', 'w'), indent=1)
PY
