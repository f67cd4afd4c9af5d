python bench.py --precision fp16x3 --steps 3 --warmup 1 --no-alt --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],4), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_classes_ms_per_step'].items()})"
