#!/bin/bash
# on-box: forward throughput against the number of hardware queues HIP may use: bash tools/hwq_sweep.sh MODEL
# (round 5: GPU_MAX_HW_QUEUES=4 costs EQTransformer 11 %, 8 and 16 equal the unset default; PhaseNet does not care)
M=${1:-phasenet}; mkdir -p gpurun_out/tmp
for q in 4 8 16; do
GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py --model $M --no-cpu-baseline --no-api --sustain-seconds 4 --detail-file gpurun_out/tmp/s_detail.json > gpurun_out/tmp/s.json 2> gpurun_out/tmp/s.err
python3 -c "
import json;d=json.loads(open('gpurun_out/tmp/s.json').read().strip().splitlines()[-1]);s=d['sustained']
print('$M GPU_MAX_HW_QUEUES=$q:', round(d['value']), 'windows/s (20-step regions)', round(d['ms_per_step']*1e3,1), 'us/step; sustained', round(s['value']), 'at', round(s['shader_clock_ghz'],3), 'GHz')" || tail -3 gpurun_out/tmp/s.err
done
