#!/bin/bash
# round-6 first look (GPU box): the headline as the driver runs it, the same loop at the per-GPU batches of a strong-scaled run (B = 32, 16, 8),
# and the launch timeline of one evaluation at B = 64 / 16 / 8.  Results: gpurun_out/r06scan/
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06scan; mkdir -p $O
cd $R
for b in 64 32 16 8; do
  python bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline --batch $b > $O/bench_b$b.json 2> $O/bench_b$b.err
done
for b in 64 16 8; do
  ( cd /tmp && export TMPDIR=/tmp
    out=$O/tl_$b
    rocprofv3 --kernel-trace --output-format csv -d $out -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-profile --batch $b > $out.log 2>&1
    t=$(find $out -name '*kernel_trace.csv' | head -1)
    python $R/tools/prof_summary.py timeline $t > $O/timeline_b$b.txt
    rm -rf $out )
done
python - <<'P'
import json,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r06scan'
for b in (64,32,16,8):
    try:
        j=json.loads(open('%s/bench_b%d.json'%(O,b)).read().strip().splitlines()[-1])
        print('B=%d value %.4f img/s  %.2f ms/step'%(b,j['value'],j['ms_per_step']))
    except Exception as e: print(b,'failed',e)
P
