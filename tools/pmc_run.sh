#!/bin/bash
# tuning aid: one PMC pass of the bench (counters only - never combined with tracing domains)
# usage: tools/pmc_run.sh <precision> <tag> <counter> [<counter> ...]
prec=$1; tag=$2; shift 2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rocprofv3 --pmc "$@" --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --precision $prec --steps 1 --warmup 0 > $out.log 2>&1
f=$(find $out -name '*counter_collection.csv' | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py counters $f conv_f16 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.txt
