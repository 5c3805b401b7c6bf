for c in 3 1; do for k in 5 10 20 40 80; do python bench.py --model phasenet --no-cpu-baseline --steps $k --warmup 5 --contexts $c 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); t=d['timing']['ms_per_step_all']; k=d['steps']; import statistics
print('contexts',d['config']['device_contexts'],'steps',k,'median total ms %.3f'%(statistics.median(t)*k), 'min %.3f'%(min(t)*k), 'win/s', round(d['value']))"; done; done
