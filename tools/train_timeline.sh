# on-box: gpurun -- 'bash tools/train_timeline.sh'  -> gpurun_out/train_timeline.txt
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/train_tl; rm -rf $O; mkdir -p $O; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/bench_train.py --batch 512 --dtype bf16 --steps 16 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cd $R; T=$(find $O -name "*kernel_trace.csv" | head -1)
python tools/train_timeline.py $T 136 full | tee gpurun_out/train_timeline.txt
rm -rf $O
