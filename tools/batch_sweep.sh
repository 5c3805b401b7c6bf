#!/bin/bash
# Throughput against the number of windows per step (= per forward launch): does one launch over 2-4 batches amortise the
# turn-around between forward kernels?  usage on the GPU box: bash tools/batch_sweep.sh
mkdir -p gpurun_out/tmp
for b in 256 512 1024 256 512 1024; do
timeout -k 10 200 python bench.py --model phasenet --batch $b --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 > gpurun_out/tmp/b.json 2>gpurun_out/tmp/b.err
python3 -c "
import json;b=json.load(open('gpurun_out/tmp/b.json'));print('batch $b  PhaseNet', round(b['value']/1e6,3),'M windows/s', round(b['ms_per_step']*1e3,1),'us/step =', round(b['ms_per_step']*1e3*256/$b,1), 'us per 256 windows; kernel', round(b['roofline']['kernel_ms']*1e3,1))"; done
for b in 512 1024; do
timeout -k 10 200 python bench.py --model eqtransformer --batch $b --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 > gpurun_out/tmp/b.json 2>gpurun_out/tmp/b.err
python3 -c "
import json;b=json.load(open('gpurun_out/tmp/b.json'));print('batch $b  EQT', round(b['value']/1e3,1),'k windows/s', round(b['ms_per_step']*1e3,1),'us/step =', round(b['ms_per_step']*1e3*256/$b,1), 'us per 256 windows')"; done
