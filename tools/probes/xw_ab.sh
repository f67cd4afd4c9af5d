#!/bin/bash
# conv_xw (Winograd F(2,3)) against conv_xp (direct) on the SR3-160 layer shapes: rocprofv3 kernel durations of tools/ff_probe.py
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/xw; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
summ() { python3 - "$1" <<'PY'
import csv, glob, sys, statistics
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
d = {}
for r in rows:
    n = r['Kernel_Name']
    if 'conv_x' in n:
        d.setdefault(n.split('(')[0], []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for n, v in d.items():
    print('   %-60s n=%3d  min %7.1f  median %7.1f  max %7.1f us' % (n[:60], len(v), min(v), statistics.median(v), max(v)))
PY
}
for only in ${SHAPES:-0 1 2 3 4 5}; do
  for lib in xw xp; do
    rm -rf /tmp/prof_$lib
    if [ $lib = xp ]; then export CSD_LIB_PATH=$R/conditional_score_diffusion_amd/libcsd_hip_tune.so CSD_XW=0; else unset CSD_LIB_PATH CSD_XW; fi
    ONLY=$only REPS=${REPS:-20} PREC=fp16x3 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$lib -o t -- python $R/tools/ff_probe.py > $O/probe_${lib}_$only.txt 2>&1
    echo "shape $only $lib: $(grep fp16x3 $O/probe_${lib}_$only.txt)"
    summ /tmp/prof_$lib
  done
done
