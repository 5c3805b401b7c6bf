#!/bin/bash
# on-box A/B: eqt_mid_kernel with one window per workgroup (plan_flags[2] = 2) against two (default), over the number of
# device contexts.  usage: bash tools/ab_mid2.sh "4 5 6"
mkdir -p gpurun_out/tmp
for c in ${1:-4 5 6}; do for f in "0,0,2" "0"; do
VOLPICK_PLAN_FLAGS="$f" timeout -k 10 200 python bench.py --model eqtransformer --no-cpu-baseline --sustain-seconds 0 --no-api --contexts $c > gpurun_out/tmp/m.json 2> gpurun_out/tmp/m.err
python3 -c "
import json;d=json.loads(open('gpurun_out/tmp/m.json').read().strip().splitlines()[-1])
k=[x for x in d['forward']['kernels'] if 'mid' in x['name']][0]
print('contexts $c flags $f:', round(d['value']), 'windows/s', round(d['ms_per_step']*1e3,1), 'us/step; mid', round(k['ms']*1e3,1), 'us')" || tail -3 gpurun_out/tmp/m.err
done; done
