#!/bin/bash
# HBM traffic of the bench's kernels: two PMC-only passes (FETCH_SIZE, WRITE_SIZE), summarised per (kernel, grid).
# usage: tools/pmc_hbm.sh <precision> <tag>
prec=$1; tag=$2
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$c
  rocprofv3 --pmc $c --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --precision $prec --steps 1 --warmup 0 --no-cpu-baseline --no-alt > $out.log 2>&1
  f=$(find $out -name '*counter_collection.csv' | head -1)
  python $GRAFT_REPO_ROOT/tools/prof_summary.py counters $f csd:: > $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$c.txt
  rm -rf $out
done
