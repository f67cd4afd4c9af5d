#!/bin/bash
# same-box A/B of two builds of the library: the headline loop alternating between them (ROUNDS times); LIBS="label=path ..."
# (a bench.py run asserts finiteness only with the default library; an alternative one is selected by CSD_LIB_PATH)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06ab; mkdir -p $O
cd $R
for r in $(seq 1 ${ROUNDS:-2}); do
  for lp in $LIBS; do
    l=${lp%%=*}; p=${lp#*=}
    for b in ${BATCHES:-64}; do
      if [ "$p" = "default" ]; then env -u CSD_LIB_PATH python bench.py --steps ${STEPS:-20} --warmup 5 --no-alt --no-cpu-baseline --no-profile --batch $b > $O/$l.$b.$r.json 2> $O/$l.$b.$r.err
      else CSD_LIB_PATH=$R/$p python bench.py --steps ${STEPS:-20} --warmup 5 --no-alt --no-cpu-baseline --no-profile --batch $b > $O/$l.$b.$r.json 2> $O/$l.$b.$r.err; fi
      python -c "
import json
j=json.loads(open('$O/$l.$b.$r.json').read().strip().splitlines()[-1]); print('$l B=$b round $r: %.4f img/s %.3f ms/step'%(j['value'],j['ms_per_step']))" 2>/dev/null || tail -3 $O/$l.$b.$r.err
    done
  done
done
