#!/bin/bash
# on-box A/B of EQTransformer plan flags through the bench line (20-step regions), alternating.  usage: bash tools/ab_flags_eqt.sh "FLAGS_A" "FLAGS_B" [ROUNDS]
mkdir -p gpurun_out/tmp
for r in $(seq 1 ${3:-2}); do for f in "$1" "$2"; do
VOLPICK_PLAN_FLAGS="$f" timeout -k 10 200 python bench.py --model eqtransformer --no-cpu-baseline --sustain-seconds 0 --no-api --detail-file gpurun_out/tmp/e_detail.json > gpurun_out/tmp/e.json 2> gpurun_out/tmp/e.err
python3 -c "
import json;d=json.load(open('gpurun_out/tmp/e_detail.json'))
print('round $r flags $f:', round(d['value']), 'windows/s', round(d['ms_per_step']*1e3,1), 'us/step;', ' '.join(str(round(k['ms']*1e3,1)) for k in d['forward']['kernels']))" || tail -3 gpurun_out/tmp/e.err
done; done
