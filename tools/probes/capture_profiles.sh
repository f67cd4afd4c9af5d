#!/bin/bash
# Round-end evidence: bench lines for the three precision modes and the rocprofv3 kernel-trace summary of
# the default bench command.  Run on the GPU box (gpurun); results land in gpurun_out/ for copying to profiles/.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/cap; mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --precision fp16 --no-alt > $O/bench_fp16.json 2>/dev/null
python bench.py --precision fp32 --steps 8 --warmup 1 --no-alt > $O/bench_fp32.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --no-alt > $O/bench_under_rocprof.json 2> $O/rocprof.err
f=$(find $O/trace -name '*kernel_stats.csv' | head -1); cp $f $O/kernel_stats.csv
t=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python $R/tools/prof_summary.py trace $t csd:: > $O/kernel_trace_summary.txt
rm -rf $O/trace
