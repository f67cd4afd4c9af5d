#!/bin/bash
# tuning aid: the launches of one CMDE-128 (BASELINE configs[2] shape, B = 64) network evaluation in order
# usage: tools/probes/timeline_cmde.sh <precision> <tag>
prec=$1; tag=$2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/tlc_$tag
rocprofv3 --kernel-trace --output-format csv -d $out -- python $GRAFT_REPO_ROOT/tools/probes/cmde_profile.py $prec > $out.log 2>&1
t=$(find $out -name '*kernel_trace.csv' | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py timeline $t > $GRAFT_REPO_ROOT/gpurun_out/timeline_cmde_$tag.txt
rm -rf $out
