#!/bin/bash
# Round-2 artifact set, on the GPU box: gpurun --timeout 1200 -- 'bash tools/r02_profile.sh TAG'
# -> gpurun_out/r02_TAG/: GPU suite, the default bench line (both models), rocprofv3 --kernel-trace --stats of the same
# command, bench --strong at N = 1, phase clocks of eqt_tail3_kernel / pn_window_kernel, the same-box A/B of the fused
# EQTransformer plan against the layer launches.  PMC traffic: tools/pmc_traffic.sh (separate call).
export TMPDIR=/tmp
T=${1:-a}; R=$PWD; O=$R/gpurun_out/r02_$T; mkdir -p $O
# a GPU step that was killed at its limit ends the script: no further GPU step after it
chk() { rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed (rc=$rc): stopping"; exit $rc; fi; }
timeout -k 10 500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
chk; tail -2 $O/pytest_gpu.txt
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
chk
cd /tmp
# one device context under the profiler: with three, kernels of different contexts share the chip and a kernel's span in
# the trace includes the time its workgroups wait for CUs another context's kernel still holds (AverageNs 2-3x the
# kernel's own duration); the bench line itself (bench.json) is the three-context run
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --contexts 1 > $O/bench_under_rocprof.json 2> $O/rocprof.err
chk
cd $R
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
rm -rf $O/prof
timeout -k 10 200 python bench.py --strong --steps 10 --warmup 2 > $O/bench_strong_n1.json 2> $O/strong.err
chk
timeout -k 10 100 python tools/tail_clock.py 10 > $O/eqt_tail_kernel_phases.txt 2>&1
chk
timeout -k 10 100 python tools/core_clock.py > $O/phasenet_window_kernel_phases.txt 2>&1
chk
tools/ab_e2e.sh 3 0 0,0,0,0,0,0,0,15 > $O/eqt_fused_vs_layer_launches_e2e.txt 2>&1
chk
timeout -k 10 200 python tools/ab_steps.py eqtransformer "0" "0,0,0,0,0,0,0,15" > $O/eqt_fused_vs_layer_launches_steps.txt 2>&1
chk
ls -la $O
# the bf16-piece kernels (encoder 1-6, ResCNN, decoder 0-3 stages 1-3, decoder tail) against their fp32-MFMA forms, same box
timeout -k 10 200 python tools/ab_steps.py eqtransformer "0" "0,0,0,0,0,0,0,496" > $O/eqt_bf16_pieces_vs_fp32_mfma_steps.txt 2>&1
chk
tools/ab_e2e.sh 2 0 0,0,0,0,0,0,0,496 > $O/eqt_bf16_pieces_vs_fp32_mfma_e2e.txt 2>&1
chk
ls -la $O
