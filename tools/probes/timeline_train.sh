#!/bin/bash
# tuning aid: rocprofv3 kernel trace of the training side bench -> the launches of one training step in order (between two Adam launches)
# usage: tools/probes/timeline_train.sh <precision> <tag>
prec=$1; tag=$2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/tlt_$tag
rocprofv3 --kernel-trace --output-format csv -d $out -- python $GRAFT_REPO_ROOT/tools/bench_train.py --precision $prec --steps 2 --warmup 2 > $out.log 2>&1
t=$(find $out -name '*kernel_trace.csv' | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py timeline $t adam_ema_kernel > $GRAFT_REPO_ROOT/gpurun_out/timeline_train_$tag.txt
rm -rf $out
