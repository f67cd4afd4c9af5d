cd $GRAFT_REPO_ROOT
for a in ${ABLS:-0 23 55 87 119}; do echo "== ABL $a"; CSD_FF_ABL=$a ONLY=${ONLY:-1,3} REPS=10 python tools/ff_probe.py 2>&1 | grep -v amdgpu; done > gpurun_out/abl.txt
