#!/bin/bash
# kernel-only durations of the block-convolution probe (rocprofv3 kernel trace): conv_xp vs conv_ff on the same shapes
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/xp; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in xp ff; do
  if [ $lib = ff ]; then export CSD_LIB_PATH=$R/conditional_score_diffusion_amd/libcsd_hip_tune.so CSD_XP=0; fi
  rm -rf $O/tr_$lib
  REPS=10 PREC=fp16x3 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$lib -- python $R/tools/ff_probe.py > $O/prof_$lib.log 2>&1
  t=$(find $O/tr_$lib -name '*kernel_trace.csv' | head -1)
  python - "$t" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
seq = []
for r in rows:
    n = r['Kernel_Name']
    if 'conv_xp' in n or 'conv_ff_kernel' in n or 'conv_fx' in n:
        seq.append((n.split('(')[0][:60], int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r.get('Grid_Size', 0)), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
# the probe runs shapes in order, 1 + 1 + REPS launches each
groups = collections.OrderedDict()
i = 0
cur = None
out = []
for n, g, us in seq:
    key = (n, g)
    if cur is None or key != cur[0] or len(cur[1]) >= 12:
        if cur: out.append(cur)
        cur = [key, []]
    cur[1].append(us)
if cur: out.append(cur)
for (n, g), v in out:
    v2 = sorted(v)[: max(1, len(v) - 2)]
    print('%-62s grid %6d  n=%2d  median %8.1f us  min %8.1f' % (n, g, len(v), sorted(v)[len(v) // 2], min(v)))
PY
  rm -rf $O/tr_$lib
done
