set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a; mkdir -p $O
cd $R
python bench.py --precision fp16x3 --no-alt --no-cpu-baseline > $O/bench_fp16x3.json 2> $O/bench_fp16x3.err
bash tools/timeline_run.sh fp16x3 r04a_fp16x3
cp gpurun_out/timeline_r04a_fp16x3.txt $O/
