#!/bin/bash
# on-box A/B under sustained load: alternates plan variants, each with the 21 x 20-step regions AND a >= 5 s region (shader clock
# from in-kernel stamps).  usage: bash tools/ab_sustained.sh MODEL "FLAGS_A" "FLAGS_B" [ROUNDS]
M=${1:-phasenet}; R=${4:-2}; mkdir -p gpurun_out/tmp
for r in $(seq 1 $R); do for f in "$2" "$3"; do
VOLPICK_PLAN_FLAGS="$f" timeout -k 10 200 python bench.py --model $M --no-cpu-baseline --no-api --sustain-seconds 5 > gpurun_out/tmp/s.json 2> gpurun_out/tmp/s.err
python3 -c "
import json;d=json.loads(open('gpurun_out/tmp/s.json').read().strip().splitlines()[-1]);s=d['sustained']
print('round $r flags $f:', round(d['value']), 'windows/s (20-step regions)', round(d['ms_per_step']*1e3,1), 'us/step; sustained', round(s['value']), 'at', round(s['shader_clock_ghz'],3), 'GHz')" || tail -3 gpurun_out/tmp/s.err
done; done
