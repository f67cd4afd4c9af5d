#!/bin/bash
# rocprofv3 --kernel-trace --stats of the cfg4 training bench -> gpurun_out/train_<tag>_{stats.csv,bench.json}
# usage: tools/probes/train_profile.sh <precision> <tag>
prec=${1:-fp32}; tag=${2:-r01}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/trainprof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/tools/bench_train.py --precision $prec --steps 4 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/train_${tag}_bench.json 2> $out.err
s=$(find $out -name '*kernel_stats.csv' | head -1)
cp $s $GRAFT_REPO_ROOT/gpurun_out/train_${tag}_stats.csv
head -40 $s
rm -rf $out
