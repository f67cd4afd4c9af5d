#!/bin/bash
# Round-3 evidence run (GPU box): the default bench line, the rocprofv3 kernel-trace summary of the same command, SQ counters of the
# kernels that run, HBM traffic (FETCH_SIZE / WRITE_SIZE in separate PMC-only passes), the launch timeline of one evaluation.
# Results: gpurun_out/r03/ (the summaries are copied to profiles/r03_* by hand).
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --no-alt --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err
f=$(find $O/trace -name '*kernel_stats.csv' | head -1); cp $f $O/kernel_stats.csv
t=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python $R/tools/prof_summary.py trace $t csd:: > $O/kernel_trace_summary.txt
rm -rf $O/trace
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sq -- python $R/bench.py --steps 1 --warmup 0 --no-alt --no-cpu-baseline > /dev/null 2>&1
f=$(find $O/sq -name '*counter_collection.csv' | head -1)
python $R/tools/prof_summary.py counters $f csd:: | grep -E "conv_ff|conv_f16_q|gn_apply16|gn_fused16|pw16|attention" > $O/pmc_sq.txt
rm -rf $O/sq
cd $R
bash tools/pmc_hbm.sh fp16f8 r03_fp16f8
python tools/hbm_traffic.py r03_fp16f8 $O/hbm_traffic_fp16f8.json > $O/hbm_traffic_fp16f8.txt
bash tools/timeline_run.sh fp16f8 r03_sampling_fp16f8
cp gpurun_out/timeline_r03_sampling_fp16f8.txt $O/ 2>/dev/null
