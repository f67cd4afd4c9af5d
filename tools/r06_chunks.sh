#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c; mkdir -p $O
cd $R
run() { # label, env...
  l=$1; shift
  env "$@" python bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline --no-profile --batch ${B:-64} > $O/$l.json 2> $O/$l.err
  python -c "
import json
j=json.loads(open('$O/$l.json').read().strip().splitlines()[-1]); print('$l: %.4f img/s %.2f ms/step'%(j['value'],j['ms_per_step']))" 2>/dev/null || tail -3 $O/$l.err
}
run k1 CSD_CHUNKS=1
run k2 CSD_CHUNKS=2
run k3 CSD_CHUNKS=3
run k4 CSD_CHUNKS=4
run k2seq CSD_CHUNKS=2 CSD_CHUNK_SEQ=1
run k4seq CSD_CHUNKS=4 CSD_CHUNK_SEQ=1
run k2s10 CSD_CHUNKS=2 CSD_CHUNK_SIDE=10
run k4s10 CSD_CHUNKS=4 CSD_CHUNK_SIDE=10
run k1b CSD_CHUNKS=1
