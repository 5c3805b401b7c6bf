#!/bin/bash
# Round-6 artifact set, on the GPU box:  gpurun --timeout 1200 -- 'bash tools/r06_profile.sh TAG'
# -> gpurun_out/r06_TAG/: GPU suite (with durations), smoke, the default bench line (both models + train + mseed objects), the
# launcher form `bench.py --gpus 2 --rehearse-gloo` called directly, rocprofv3 --kernel-trace --stats of the bench command (one
# context), bench --strong at N = 1, phase clocks, SQ counters of every forward kernel, the LDS bank-conflict owner pass, the HBM
# traffic passes and the training step's kernel stats.  Copy what is to be judged into profiles/.
export TMPDIR=/tmp
T=${1:-z}; R=$PWD; O=$R/gpurun_out/r06_$T; mkdir -p $O
chk() { rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed (rc=$rc): stopping"; exit $rc; fi; }
timeout -k 10 600 python -m pytest tests -m gpu -q --durations=15 > $O/pytest_gpu.txt 2>&1
chk; tail -2 $O/pytest_gpu.txt
timeout -k 10 200 python __graft_entry__.py smoke > $O/smoke.txt 2>&1
chk; tail -1 $O/smoke.txt
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 --detail-file $O/bench_detail.json > $O/bench.json 2> $O/bench.err
chk; echo "bench done"
timeout -k 10 300 python bench.py --gpus 2 --rehearse-gloo --steps 20 --warmup 5 --detail-file $O/bench_gpus2_rehearsal_detail.json > $O/bench_gpus2_rehearsal.json 2> $O/bench_gpus2_rehearsal.err
chk; echo "launcher rehearsal done"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0 --contexts 1 --detail-file $O/bench_under_rocprof_detail.json > $O/bench_under_rocprof.json 2> $O/rocprof.err
chk
cd $R
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
rm -rf $O/prof; echo "kernel stats done"
timeout -k 10 200 python bench.py --strong --steps 10 --warmup 2 --detail-file $O/bench_strong_n1_detail.json > $O/bench_strong_n1.json 2> $O/strong.err
chk
timeout -k 10 100 python tools/tail_clock.py 10 > $O/eqt_tail_kernel_phases.txt 2>&1
chk
timeout -k 10 100 python tools/mid_clock.py > $O/eqt_mid_kernel_phases.txt 2>&1
chk
timeout -k 10 100 python tools/core_clock.py > $O/phasenet_window_kernel_phases.txt 2>&1
chk; echo "clocks done"
bash tools/pmc_sq_all.sh r06_$T 4 > $O/sq.log 2>&1
chk; cp gpurun_out/sq_r06_$T/summary.txt $O/sq_counters_summary.txt; echo "sq done"
bash tools/pmc_lds_owner.sh > $O/lds_owner.log 2>&1
chk; cp gpurun_out/lds_owner/summary.txt $O/lds_bank_conflict_owner.txt; echo "lds owner done"
TRAFFIC_JSON=r06_traffic.json bash tools/pmc_traffic.sh > $O/traffic.log 2>&1
chk; cp profiles/r06_traffic.json $O/traffic.json 2>/dev/null; tail -12 $O/traffic.log
bash tools/train_profile.sh bf16 > $O/train_bf16_profile.txt 2>&1
chk; cp gpurun_out/train_prof_bf16/kernel_stats.csv $O/train_bf16_b512_kernel_stats.csv 2>/dev/null
ls $O
