#!/bin/bash
# samples rocm-smi (socket power, shader clock, temperature) twice a second while the default bench runs and while the MFMA probe runs:
# the evidence behind DESIGN.md's "power-bound" paragraph.  usage: tools/power_trace.sh <tag>
tag=${1:-r04}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/power_$tag.txt
cd $R
rocm-smi --showmaxpower 2>/dev/null | grep -iE "GPU\[" > $O; rocm-smi --showpowercap 2>/dev/null | grep -iE "GPU\[" >> $O 2>/dev/null
echo "--- idle" >> $O
rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -iE "sclk|Socket|power \(W\)|junction" >> $O
python bench.py --steps 300 --warmup 3 --no-alt --no-cpu-baseline --no-profile > gpurun_out/power_bench_$tag.json 2>/dev/null &
pid=$!
echo "--- during python bench.py (fp16x3, B = 64): one sample per 0.5 s" >> $O
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -iE "sclk|power \(W\)" | tr '\n' ' ' >> $O; echo >> $O
  sleep 0.5
done
tail -1 gpurun_out/power_bench_$tag.json | cut -c1-200 >> $O
sample() {   # $1 = pid to follow
  while kill -0 $1 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -iE "sclk|power \(W\)" | tr '\n' ' ' >> $O; echo >> $O
    sleep 0.3
  done
}
echo "--- during 8000 back-to-back conv_xk launches (160^2, 96 -> 96, B = 64: tools/ff_probe.py ONLY=0)" >> $O
ONLY=0 REPS=8000 PREC=fp16x3 python tools/ff_probe.py > gpurun_out/power_ffprobe_$tag.txt 2>/dev/null &
sample $!
grep fp16x3 gpurun_out/power_ffprobe_$tag.txt >> $O
if [ -x build/xp_order_probe ]; then
  echo "--- during tools/xp_order_probe long: MFMA-only stream, RANDOM fp16 operands (3500 launches), then CONSTANT operands (5000)" >> $O
  ./build/xp_order_probe long > gpurun_out/power_probe_$tag.txt &
  sample $!
  cat gpurun_out/power_probe_$tag.txt >> $O
fi
