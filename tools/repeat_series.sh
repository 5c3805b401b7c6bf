# on-box: the bench's per-repeat rates in order (k windows/s), many repeats of the driver's 20-step region
for m in phasenet eqtransformer; do
python bench.py --model $m --no-cpu-baseline --steps 20 --warmup 5 --repeats ${1:-60} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$m median', round(d['value']))
print([round(256/x) for x in d['timing']['ms_per_step_all']])
"
done
