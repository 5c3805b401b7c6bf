#!/bin/bash
# on-box: kernel trace of a short EQTransformer bench run -> gpurun_out/pipe_trace/<tag>_kernel_trace.csv (start / end of every launch)
# usage: [M=phasenet] tools/pipe_trace.sh TAG CONTEXTS [PLAN_FLAGS]
export TMPDIR=/tmp
T=$1; N=$2; F=${3:-0}
R=$PWD; O=$R/gpurun_out/pipe_trace; mkdir -p $O; cd /tmp
VOLPICK_PLAN_FLAGS="$F" rocprofv3 --kernel-trace --output-format csv -d $O/$T -- python3 $R/bench.py --model ${M:-eqtransformer} --no-cpu-baseline --steps 12 --warmup 3 --repeats 1 --contexts $N > $O/$T.json 2> $O/$T.err
cd $R
find $O/$T -name "*kernel_trace.csv" -exec cp {} $O/${T}_kernel_trace.csv \;
rm -rf $O/$T
