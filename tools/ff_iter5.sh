#!/bin/bash
cd $GRAFT_REPO_ROOT
for a in 0 7 39 103 0; do echo "== persistent ABL $a"; CSD_FF_ABL=$a ONLY=1 REPS=10 PREC=fp16x3 python tools/ff_probe.py 2>&1 | grep -v amdgpu; done > gpurun_out/iter5.txt
