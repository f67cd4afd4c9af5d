#!/bin/bash
# tuning aid: the launches of one NCSN++ (SR3-160 hyper-parameters, B = 64) evaluation in order
# usage: tools/probes/timeline_ncsnpp.sh <precision> <tag>
prec=$1; tag=$2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/tln_$tag
rocprofv3 --kernel-trace --output-format csv -d $out -- python $GRAFT_REPO_ROOT/tools/probes/ncsnpp_profile.py $prec > $out.log 2>&1
t=$(find $out -name '*kernel_trace.csv' | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py timeline $t > $GRAFT_REPO_ROOT/gpurun_out/timeline_ncsnpp_$tag.txt
rm -rf $out
