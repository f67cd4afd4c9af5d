#!/bin/bash
# SQ counters of conv_xw / conv_xp in isolation (tools/ff_probe.py shape $1): one --pmc pass each (counters only)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/xw; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in xw xp; do
  if [ $lib = xp ]; then export CSD_LIB_PATH=$R/conditional_score_diffusion_amd/libcsd_hip_tune.so CSD_XW=0; else unset CSD_LIB_PATH CSD_XW; fi
  rm -rf /tmp/pmc_$lib
  ONLY=${1:-0} REPS=4 PREC=fp16x3 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc_$lib -- python $R/tools/ff_probe.py > /dev/null 2>&1
  f=$(find /tmp/pmc_$lib -name '*counter_collection.csv' | head -1)
  python $R/tools/prof_summary.py counters $f conv_x | tee $O/pmc_${lib}_${1:-0}.txt
done
