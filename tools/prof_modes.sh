#!/bin/bash
# rocprofv3 kernel trace of a short bench in the fp16x3 and fp16 modes -> gpurun_out/prof_<mode>_{stats.csv,summary.txt}
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
STEPS=${1:-3}
cd /tmp && export TMPDIR=/tmp
for mode in fp16x3 fp16; do
  rm -rf /tmp/prof_$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -- python $R/bench.py --steps $STEPS --warmup 1 --no-alt --no-cpu-baseline --no-profile --precision $mode > $O/prof_${mode}_bench.json 2> $O/prof_${mode}.err
  f=$(find /tmp/prof_$mode -name '*kernel_stats.csv' | head -1); cp $f $O/prof_${mode}_stats.csv
  t=$(find /tmp/prof_$mode -name '*kernel_trace.csv' | head -1)
  python $R/tools/prof_summary.py trace $t csd:: > $O/prof_${mode}_summary.txt
done
