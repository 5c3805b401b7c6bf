export TMPDIR=/tmp
mkdir -p gpurun_out/r01e
python tools/bench_train.py --batch 512 --torch-gpu 2>/dev/null | tail -1 > gpurun_out/r01e/train_b512_bench.json
python tools/bench_train.py --batch 1024 --torch-gpu --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r01e/train_b1024_bench.json
R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01e/prof -- python3 $R/tools/bench_train.py --batch 512 --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r01e/train_b512_bench_under_rocprof.json 2>/dev/null
cd $R; find gpurun_out/r01e -name "*_kernel_trace.csv" -delete; find gpurun_out/r01e -name "*.db" -delete
python tools/bench_mseed.py 2>/dev/null | tail -1 > gpurun_out/r01e/mseed_reclen4096_bench.json
python tools/bench_mseed.py --reclen 512 2>/dev/null | tail -1 > gpurun_out/r01e/mseed_reclen512_bench.json
ls -R gpurun_out/r01e | head -20
