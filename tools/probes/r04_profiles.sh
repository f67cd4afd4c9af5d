#!/bin/bash
# Round-4 evidence run (GPU box): the default bench line (fp16x3), the rocprofv3 kernel-trace summary of the same command, SQ counters of
# the kernels that run, HBM traffic (FETCH_SIZE / WRITE_SIZE in separate PMC-only passes), the launch timeline of one evaluation, the
# CPU baseline's thread sweep.  Results: gpurun_out/r04/ (the summaries are copied to profiles/r04_* by hand).
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
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
python $R/tools/prof_summary.py counters $f csd:: | grep -E "conv_xp|conv_ff|conv_f16_q|gn_apply16|gn_fused16|pw16|attention" > $O/pmc_sq.txt
rm -rf $O/sq
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/grbm -- python $R/bench.py --steps 1 --warmup 0 --no-alt --no-cpu-baseline > /dev/null 2>&1
f=$(find $O/grbm -name '*counter_collection.csv' | head -1)
python $R/tools/prof_summary.py counters $f csd:: | grep -E "conv_xp|conv_f16_q" > $O/pmc_grbm.txt
rm -rf $O/grbm
cd $R
bash tools/pmc_hbm.sh fp16x3 r04_fp16x3
python tools/hbm_traffic.py r04_fp16x3 $O/hbm_traffic_fp16x3.json > $O/hbm_traffic_fp16x3.txt
bash tools/timeline_run.sh fp16x3 r04_sampling_fp16x3
cp gpurun_out/timeline_r04_sampling_fp16x3.txt $O/ 2>/dev/null
# the training step and the NCSN++ forwards (kernel stats of the side benches)
bash tools/train_prof.sh ddpm_paired r04 > $O/train_bench_ddpm_paired.json; cp gpurun_out/train_r04_kernel_stats.csv $O/train_ddpm_paired_kernel_stats.csv
python tools/bench_train.py --model ncsnpp_paired --precision fp16x3 2>/dev/null | tail -1 > $O/train_bench_ncsnpp_paired.json
python tools/bench_train.py --model ddpm_paired --precision fp16x3 --batch 7 2>/dev/null | tail -1 > $O/train_bench_ddpm_paired_b7.json
bash tools/ncsnpp_timeline.sh 160 r04_160 > /dev/null; cp gpurun_out/ncsnpp_r04_160_kernel_stats.csv $O/ncsnpp160_kernel_stats.csv
bash tools/ncsnpp_timeline.sh 256 r04_256 > /dev/null; cp gpurun_out/ncsnpp_r04_256_kernel_stats.csv $O/ncsnpp256_kernel_stats.csv
if [ "${1:-}" = "sweep" ]; then python bench.py --cpu-thread-sweep > $O/cpu_thread_sweep.txt 2>&1; fi
