#!/bin/bash
# tuning aid: rocprofv3 kernel trace of a short bench run -> the launches of one network evaluation in order
# usage: tools/timeline_run.sh <precision> <tag>
prec=$1; tag=$2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/tl_$tag
rocprofv3 --kernel-trace --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --precision $prec --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-profile > $out.log 2>&1
t=$(find $out -name '*kernel_trace.csv' | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py timeline $t > $GRAFT_REPO_ROOT/gpurun_out/timeline_$tag.txt
rm -rf $out
