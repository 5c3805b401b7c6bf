# on-box: bench lines against the number of device contexts and of HIP hardware queues (GPU_MAX_HW_QUEUES, read when the
# HIP runtime starts).  usage: bash tools/ctx_sweep.sh MODEL STEPS ROUNDS "Q:C Q:C ..."
M=${1:-eqtransformer}; S=${2:-50}; R=${3:-3}; L=${4:-"4:3 8:3 8:4 8:5 8:6"}
for r in $(seq 1 $R); do for qc in $L; do q=${qc%%:*}; c=${qc##*:}
echo -n "round $r hwq=$q contexts=$c $M steps=$S: "; GPU_MAX_HW_QUEUES=$q python bench.py --model $M --no-cpu-baseline --steps $S --contexts $c 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4), 'min/max', round(d['timing']['windows_per_s_min']), round(d['timing']['windows_per_s_max']))"
done; done
