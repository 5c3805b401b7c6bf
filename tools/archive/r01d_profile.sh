set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/r01d
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r01d/pytest_gpu.txt
python bench.py --steps 200 --warmup 20 > gpurun_out/r01d/pn_bench.json 2> gpurun_out/r01d/pn_bench.err
python bench.py --model eqtransformer --steps 100 --warmup 10 > gpurun_out/r01d/eqt_bench.json 2> gpurun_out/r01d/eqt_bench.err
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01d/pn_prof -- python3 $R/bench.py --steps 60 --warmup 6 --no-cpu-baseline --contexts 1 > $R/gpurun_out/r01d/pn_bench_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01d/eqt_prof -- python3 $R/bench.py --model eqtransformer --steps 40 --warmup 4 --no-cpu-baseline --contexts 1 > $R/gpurun_out/r01d/eqt_bench_rocprof.json 2>/dev/null
cd $R
find gpurun_out/r01d -name "*kernel_stats.csv" | head
find gpurun_out/r01d -name "*_kernel_trace.csv" -delete
find gpurun_out/r01d -name "*.db" -delete
cat gpurun_out/r01d/pytest_gpu.txt
tail -c 1500 gpurun_out/r01d/pn_bench.json
tail -c 1500 gpurun_out/r01d/eqt_bench.json
