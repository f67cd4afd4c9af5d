#!/bin/bash
# Round-2 evidence run (GPU box): bench lines, rocprofv3 kernel-trace summary of the default bench command, SQ counters of the kernels
# that actually run (conv_ff / conv_f16_q / gn_apply16 / pw16), HBM traffic (FETCH_SIZE / WRITE_SIZE in separate PMC-only passes).
# Results: gpurun_out/r02/ (copied to profiles/ by hand).
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --precision fp16 --no-alt --no-cpu-baseline > $O/bench_fp16.json 2>/dev/null
python bench.py --precision fp16x3 --no-alt --no-cpu-baseline > $O/bench_fp16x3.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --no-alt --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err
f=$(find $O/trace -name '*kernel_stats.csv' | head -1); cp $f $O/kernel_stats.csv
t=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python $R/tools/prof_summary.py trace $t csd:: > $O/kernel_trace_summary.txt
rm -rf $O/trace
# SQ counters (one PMC-only pass, 8 counters)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sq -- python $R/bench.py --steps 1 --warmup 0 --no-alt --no-cpu-baseline > /dev/null 2>&1
f=$(find $O/sq -name '*counter_collection.csv' | head -1)
python $R/tools/prof_summary.py counters $f csd:: | grep -E "conv_ff|conv_f16_q|gn_apply16|pw16|attention" > $O/pmc_sq.txt
rm -rf $O/sq
cd $R
for mode in fp16f8 fp16x3; do
  bash tools/pmc_hbm.sh $mode r02_$mode
  python tools/hbm_traffic.py r02_$mode $O/hbm_traffic_$mode.json > $O/hbm_traffic_$mode.txt
done
