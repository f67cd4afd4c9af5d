#!/bin/bash
# round-6 quick loop (GPU box): parity tests (TESTS, default the network-level files) + the headline at the batches in BATCHES + an
# optional launch timeline (TIMELINE=1) under gpurun_out/r06q/
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06q; mkdir -p $O
cd $R
if [ "${TESTS:-net}" = "net" ]; then T="tests/test_gpu_fullsize.py tests/test_gpu_network.py tests/test_gpu_steps.py"; elif [ "$TESTS" = "none" ]; then T=""; else T="$TESTS"; fi
if [ -n "$T" ]; then
  timeout 2400 python -m pytest $T -x -q -m gpu 2>&1 | tail -8 > $O/tests.txt
  cat $O/tests.txt
fi
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
if [ "${TIMELINE:-0}" = "1" ]; then
  for b in ${BATCHES:-64}; do
  ( cd /tmp && export TMPDIR=/tmp
    out=$O/tl_$b
    rocprofv3 --kernel-trace --output-format csv -d $out -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-profile --batch $b > $out.log 2>&1
    t=$(find $out -name '*kernel_trace.csv' | head -1)
    python $R/tools/prof_summary.py timeline $t > $O/timeline_b$b.txt
    rm -rf $out )
  done
fi
