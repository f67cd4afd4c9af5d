#!/bin/bash
# rocprofv3 kernel trace of NCSN++ forwards (SR3-160 hyper-parameters, B = 64 | NCSN++-256 B = 8) -> per-kernel stats of the run
# usage: tools/ncsnpp_timeline.sh <160|256> <tag>
which=${1:-160}; tag=${2:-ncsnpp}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/np_$tag
cat > /tmp/np_run.py <<PY
import sys
sys.path.insert(0, '$GRAFT_REPO_ROOT'); sys.path.insert(0, '$GRAFT_REPO_ROOT/tools')
sys.argv = ['bench_other.py', 'fp16x3', 'import']
import bench_other as bo
if '$which' == '160':
    bo.ncsnpp('ncsnpp_paired', 64, reps=3)
else:
    bo.ncsnpp256(8, reps=3)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python /tmp/np_run.py > $out.log 2>&1
cp $(find $out -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/ncsnpp_${tag}_kernel_stats.csv
tail -2 $out.log | cut -c1-300
rm -rf $out
