#!/bin/bash
# raw rocprofv3 kernel trace of a 2-step bench run -> gpurun_out/r06trace/kernel_trace_b$B.csv (analysed offline: stream overlap of the chunk region)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for b in ${BATCHES:-64}; do
  out=$O/tl_$b
  rocprofv3 --kernel-trace --output-format csv -d $out -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-profile --batch $b > $out.log 2>&1
  t=$(find $out -name '*kernel_trace.csv' | head -1)
  cp $t $O/kernel_trace_b$b.csv
  rm -rf $out
done
