#!/bin/bash
# SQ counters + kernel durations of the fused-prologue block convolution in isolation (tools/ff_probe.py): one --pmc pass (8 SQ
# counters, never combined with tracing), one --kernel-trace pass -> gpurun_out/ff_{pmc,trace}.txt
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export REPS=${REPS:-4}
python $R/tools/ff_probe.py > $O/ff_probe.txt 2>&1
rm -rf /tmp/ffpmc /tmp/fftr
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/ffpmc -- python $R/tools/ff_probe.py > /dev/null 2>&1
f=$(find /tmp/ffpmc -name '*counter_collection.csv' | head -1)
python $R/tools/prof_summary.py counters $f conv_ff > $O/ff_pmc.txt
rocprofv3 --kernel-trace --output-format csv -d /tmp/fftr -- python $R/tools/ff_probe.py > /dev/null 2>&1
t=$(find /tmp/fftr -name '*kernel_trace.csv' | head -1)
python $R/tools/prof_summary.py trace $t conv_ff > $O/ff_trace.txt
