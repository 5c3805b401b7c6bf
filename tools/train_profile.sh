export TMPDIR=/tmp
python tools/bench_train.py --batch 512 2>&1 | tail -1
python tools/bench_train.py --batch 1024 --no-cpu-baseline 2>&1 | tail -1
R=$PWD; mkdir -p gpurun_out/train_prof; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/train_prof -- python3 $R/tools/bench_train.py --batch 512 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $R; find gpurun_out/train_prof -name "*_kernel_trace.csv" -delete; find gpurun_out/train_prof -name "*.db" -delete
head -40 gpurun_out/train_prof/*/*kernel_stats.csv | cut -c1-220
