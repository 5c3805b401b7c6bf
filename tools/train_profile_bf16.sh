export TMPDIR=/tmp
R=$PWD; mkdir -p gpurun_out/train_prof_bf16; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/train_prof_bf16 -- python3 $R/tools/bench_train.py --batch 512 --steps 10 --warmup 2 --no-cpu-baseline --dtype bf16 > /dev/null 2>&1
cd $R; find gpurun_out/train_prof_bf16 -name "*_kernel_trace.csv" -delete; find gpurun_out/train_prof_bf16 -name "*.db" -delete
head -30 gpurun_out/train_prof_bf16/*/*kernel_stats.csv | cut -c1-170
