#!/bin/bash
# tuning aid: rocprofv3 kernel trace of a short bench run, summarised per (kernel, grid)
# usage: tools/trace_run.sh <precision> <tag> [filter]
prec=$1; tag=$2; filt=${3:-csd::}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/trace_$tag
rocprofv3 --kernel-trace --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --precision $prec --steps 2 --warmup 1 --no-cpu-baseline --no-alt > $out.log 2>&1
t=$(find $out -name '*kernel_trace.csv' | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py trace $t $filt > $GRAFT_REPO_ROOT/gpurun_out/trace_$tag.txt
rm -rf $out
