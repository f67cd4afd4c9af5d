#!/bin/bash
# A/B of conv_xp variants (libcsd_hip_<name>.so built from different conv_xp.hip states): kernel-only medians of the block-conv probe
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/xp; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  export CSD_LIB_PATH=$R/conditional_score_diffusion_amd/libcsd_hip_$lib.so
  rm -rf $O/tr_$lib
  ONLY=1,2,3 REPS=10 PREC=fp16x3 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$lib -- python $R/tools/ff_probe.py > $O/prof_$lib.log 2>&1
  t=$(find $O/tr_$lib -name '*kernel_trace.csv' | head -1)
  echo "== $lib"
  python - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
cur, out = None, []
for r in rows:
    n = r['Kernel_Name']
    if 'conv_xp' not in n: continue
    us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    key = n.split('(')[0][-34:]
    if cur is None or key != cur[0] or len(cur[1]) >= 12:
        if cur: out.append(cur)
        cur = [key, []]
    cur[1].append(us)
if cur: out.append(cur)
for k, v in out:
    print('  %-36s n=%2d median %8.1f min %8.1f' % (k, len(v), sorted(v)[len(v) // 2], min(v)))
PY
  rm -rf $O/tr_$lib
done
