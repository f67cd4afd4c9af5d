#!/bin/bash
# tuning aid: kernel-busy time vs wall of the bench loop without the event profiler (launch gaps)
prec=$1
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/gap_$prec
rocprofv3 --kernel-trace --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --precision $prec --steps 6 --warmup 2 --no-alt --no-cpu-baseline --no-profile > $out.log 2>&1
t=$(find $out -name '*kernel_trace.csv' | head -1)
python - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if 'csd' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last 4 PC steps' worth: find sampler update kernels as step markers
marks = [i for i, r in enumerate(rows) if 'reverse_diffusion_update' in r['Kernel_Name']]
a, b = marks[-5], marks[-1]
seg = rows[a + 1:b + 1]
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
wall = int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])
print('4 steps: wall %.2f ms/step, kernel-busy %.2f ms/step, gaps %.2f ms/step, %d launches/step' % (wall / 4e6, busy / 4e6, (wall - busy) / 4e6, len(seg) // 4))
PY
rm -rf $out
