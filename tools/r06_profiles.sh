#!/bin/bash
# Round-6 evidence run (GPU box): the default bench line (fp16x3), the rocprofv3 kernel-trace summary of the same command, SQ counters of
# the kernels that run (MFMA-busy of BOTH conv_xk variants), HBM traffic (FETCH_SIZE / WRITE_SIZE in separate PMC-only passes), the
# launch timeline of one evaluation at B = 64 and B = 8, the training step and the NCSN++ forwards.
# Results: gpurun_out/r06/ (the summaries are copied to profiles/r06_* by hand).
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-alt --no-cpu-baseline > $O/bench_as_driver.json 2> $O/bench_as_driver.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --no-alt --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err
f=$(find $O/trace -name '*kernel_stats.csv' | head -1); cp $f $O/kernel_stats.csv
t=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python $R/tools/prof_summary.py trace $t csd:: > $O/kernel_trace_summary.txt
rm -rf $O/trace
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sq -- python $R/bench.py --steps 1 --warmup 0 --no-alt --no-cpu-baseline > /dev/null 2>&1
f=$(find $O/sq -name '*counter_collection.csv' | head -1)
python $R/tools/prof_summary.py counters $f csd:: | grep -E "conv_xk|conv_f16_q|gn_apply16|gn_fused16|pw16_kernel|attention" > $O/pmc_sq.txt
python - $O/pmc_sq.txt >> $O/pmc_sq.txt <<'P'
import re, sys, collections
d = collections.defaultdict(dict)
for l in open(sys.argv[1]):
    m = re.match(r'(\S.*?)\s+wgs=(\d+)\s+(\w+)\s+avg=([\d.e+]+) \(n=(\d+)\)', l.strip())
    if m and m.group(1).startswith('conv_xk'):
        d[m.group(1)][m.group(3)] = (float(m.group(4)), int(m.group(5)))
print('# MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES) per conv_xk variant (launch-count weighted over its grids)')
tb = tw = 0.0
for k, v in sorted(d.items()):
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in v and 'SQ_WAVE_CYCLES' in v:
        b, n = v['SQ_VALU_MFMA_BUSY_CYCLES']; w, _ = v['SQ_WAVE_CYCLES']
        print('# %-40s launches %3d  mfma_busy %.3f' % (k, n, b / (4 * w)))
        tb += b * n; tw += w * n
if tw: print('# all conv_xk variants combined: mfma_busy %.3f' % (tb / (4 * tw)))
P
rm -rf $O/sq
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/grbm -- python $R/bench.py --steps 1 --warmup 0 --no-alt --no-cpu-baseline > /dev/null 2>&1
f=$(find $O/grbm -name '*counter_collection.csv' | head -1)
python $R/tools/prof_summary.py counters $f csd:: | grep -E "conv_xk|conv_f16_q" > $O/pmc_grbm.txt
rm -rf $O/grbm
cd $R
bash tools/pmc_hbm.sh fp16x3 r06_fp16x3
python tools/hbm_traffic.py r06_fp16x3 $O/hbm_traffic_fp16x3.json > $O/hbm_traffic_fp16x3.txt
for b in 64 8; do
  ( cd /tmp && out=$O/tl_$b
    rocprofv3 --kernel-trace --output-format csv -d $out -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-profile --batch $b > $out.log 2>&1
    t=$(find $out -name '*kernel_trace.csv' | head -1)
    python $R/tools/prof_summary.py timeline $t > $O/timeline_sampling_fp16x3_b$b.txt
    rm -rf $out $out.log )
done
# the training step and the NCSN++ forwards (kernel stats of the side benches)
bash tools/train_prof.sh ddpm_paired r06 > $O/train_bench_ddpm_paired.json; cp gpurun_out/train_r06_kernel_stats.csv $O/train_ddpm_paired_kernel_stats.csv
python tools/bench_train.py --model ddpm_paired --precision fp16x3 2>/dev/null | tail -1 > $O/train_bench_ddpm_paired_plain.json
python tools/bench_train.py --model ncsnpp_paired --precision fp16x3 2>/dev/null | tail -1 > $O/train_bench_ncsnpp_paired.json
bash tools/train_prof_b7.sh > $O/train_b7.txt 2>&1; cp gpurun_out/train_b7_kernel_stats.csv $O/train_ddpm_paired_b7_kernel_stats.csv
bash tools/ncsnpp_timeline.sh 160 r06_160 > $O/ncsnpp160.txt; cp gpurun_out/ncsnpp_r06_160_kernel_stats.csv $O/ncsnpp160_kernel_stats.csv
bash tools/ncsnpp_timeline.sh 256 r06_256 > $O/ncsnpp256.txt; cp gpurun_out/ncsnpp_r06_256_kernel_stats.csv $O/ncsnpp256_kernel_stats.csv
python tools/bench_other.py fp16x3 bench > $O/side_benches.txt 2>&1
if [ "${1:-}" = "sweep" ]; then python bench.py --cpu-thread-sweep > $O/cpu_thread_sweep.txt 2>&1; fi
