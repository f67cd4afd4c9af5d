#!/bin/bash
# is the B = 7 training step (one rank's share of the global batch of 50 on 8 GPUs) bound by the host's launch rate or by the GPU?
# kernel-time sum and launch count from rocprofv3 next to the step time without the profiler
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/tp_b7
python $R/tools/bench_train.py --model ddpm_paired --precision fp16x3 --batch 7 --steps 20 --warmup 5 2>/dev/null | grep "^{" | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $R/tools/bench_train.py --model ddpm_paired --precision fp16x3 --batch 7 --steps 10 --warmup 3 > $out.log 2>&1
t=$(find $out -name '*kernel_stats.csv' | head -1)
cp $t $R/gpurun_out/train_b7_kernel_stats.csv
grep "^{" $out.log | tail -1
python3 - $t <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows); n = sum(int(r['Calls']) for r in rows)
print('13 steps: %d launches (%.0f per step), kernel time %.2f ms per step, mean %.2f us per launch' % (n, n / 13, tot / 13e6, tot / n / 1e3))
PY
rm -rf $out
