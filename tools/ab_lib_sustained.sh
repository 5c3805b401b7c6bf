#!/bin/bash
# on-box A/B of two BUILDS of the library under sustained load: bash tools/ab_lib_sustained.sh MODEL ROUNDS LIB_A LIB_B  (paths of libvolpick_hip.so builds)
M=${1:-eqtransformer}; R=${2:-2}; shift 2; mkdir -p gpurun_out/tmp
for r in $(seq 1 $R); do for L in "$@"; do
VOLPICK_HIP_LIB="$PWD/$L" timeout -k 10 200 python bench.py --model $M --no-cpu-baseline --no-api --sustain-seconds 4 --detail-file gpurun_out/tmp/s_detail.json > gpurun_out/tmp/s.json 2> gpurun_out/tmp/s.err
python3 -c "
import json;d=json.loads(open('gpurun_out/tmp/s.json').read().strip().splitlines()[-1]);s=d['sustained']
print('round $r lib $L:', round(d['value']), 'windows/s (20-step regions)', round(d['ms_per_step']*1e3,1), 'us/step; sustained', round(s['value']), 'at', round(s['shader_clock_ghz'],3), 'GHz; kernel_ms', d['roofline']['kernel_ms'])" || tail -3 gpurun_out/tmp/s.err
done; done
