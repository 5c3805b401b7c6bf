# Final round-1 artifacts: full GPU suite, bench lines (inference both models, training), rocprofv3 kernel stats.
set -x
export TMPDIR=/tmp
O=gpurun_out/r01f; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/pytest_gpu.txt
python bench.py --steps 200 --warmup 20 > $O/pn_bench.json 2>/dev/null
python bench.py --model eqtransformer --steps 100 --warmup 10 > $O/eqt_bench.json 2>/dev/null
python tools/bench_train.py --batch 512 --torch-gpu 2>/dev/null | tail -1 > $O/train_b512_bench.json
python tools/bench_train.py --batch 1024 --torch-gpu --no-cpu-baseline 2>/dev/null | tail -1 > $O/train_b1024_bench.json
python tools/bench_mseed.py 2>/dev/null | tail -1 > $O/mseed_reclen4096_bench.json
python tools/multi_station.py 2>/dev/null | tail -4 > $O/multi_station.txt
python tools/long_stream.py 2>/dev/null | tail -8 > $O/long_stream.txt
R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/pn_prof -- python3 $R/bench.py --steps 60 --warmup 6 --no-cpu-baseline --contexts 1 > $R/$O/pn_bench_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/eqt_prof -- python3 $R/bench.py --model eqtransformer --steps 40 --warmup 4 --no-cpu-baseline --contexts 1 > $R/$O/eqt_bench_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/train_prof -- python3 $R/tools/bench_train.py --batch 512 --steps 20 --warmup 3 --no-cpu-baseline > $R/$O/train_bench_rocprof.json 2>/dev/null
cd $R
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*.db" -delete
cat $O/pytest_gpu.txt; cat $O/multi_station.txt; cat $O/long_stream.txt
