for p in fp16 fp16x3; do python bench.py --precision $p --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print(d['config'].get('precision'), round(d['value'],4), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_classes_ms_per_step'].items()})"; done
