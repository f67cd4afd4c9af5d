#!/bin/bash
# rocprofv3 kernel durations of tools/ff_probe.py shape $SHAPE for the tuning library and every ablation library given (XW_ABL values)
set -u
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for n in "$@"; do
  if [ $n = full ]; then lib=$R/conditional_score_diffusion_amd/libcsd_hip_tune.so; else lib=$R/conditional_score_diffusion_amd/libcsd_hip_xwabl$n.so; fi
  rm -rf /tmp/prof_a
  CSD_LIB_PATH=$lib ONLY=${SHAPE:-0} REPS=${REPS:-20} PREC=fp16x3 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_a -o t -- python $R/tools/ff_probe.py > /dev/null 2>&1
  python3 - $n <<'PY'
import csv, glob, sys, statistics
f = glob.glob('/tmp/prof_a/**/*kernel_trace.csv', recursive=True)
v = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f[0])) if 'conv_x' in r['Kernel_Name']]
print('ABL %-5s n=%2d  min %7.1f  median %7.1f us' % (sys.argv[1], len(v), min(v), statistics.median(v)))
PY
done
