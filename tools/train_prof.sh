#!/bin/bash
# rocprofv3 kernel stats of the training step (tools/bench_train.py): usage tools/train_prof.sh <model> <tag>
model=${1:-ddpm_paired}; tag=${2:-train}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/tp_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/tools/bench_train.py --model $model --precision fp16x3 --steps 10 --warmup 3 > $out.log 2>&1
t=$(find $out -name '*kernel_stats.csv' | head -1)
cp $t $GRAFT_REPO_ROOT/gpurun_out/train_${tag}_kernel_stats.csv
grep "^{" $out.log | tail -1
rm -rf $out
