#!/bin/bash
# tuning aid: PMC passes (counters only) over tools/wgrad_probe.py for both weight-gradient schedules
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
shape="50 128 128 64 6"
for mode in wide narrow; do
  if [ $mode = narrow ]; then export CSD_WGRAD_NARROW=1; else unset CSD_WGRAD_NARROW; fi
  python $R/tools/wgrad_probe.py $shape
  i=0
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD" \
             "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
             "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
             "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
    i=$((i+1))
    out=$R/gpurun_out/wpmc_${mode}_$i
    rocprofv3 --pmc $set --output-format csv -d $out -- python $R/tools/wgrad_probe.py $shape > $out.log 2>&1
    f=$(find $out -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python - "$f" $mode <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    if 'wgrad' not in k or 'reduce' in k: continue
    acc[k[:40]][r['Counter_Name']] += float(r['Counter_Value']); n[(k[:40], r['Counter_Name'])] += 1
for k, d in acc.items():
    print(sys.argv[2], k, {c: '%.4g' % (v / n[(k, c)]) for c, v in d.items()})
PY
    [ -z "$f" ] && tail -3 $out.log
    rm -rf $out
  done
done
