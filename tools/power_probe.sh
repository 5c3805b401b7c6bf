#!/bin/bash
# on-box: socket power and shader clock (rocm-smi) sampled while one model's bench loop runs for a few seconds
# usage: tools/power_probe.sh phasenet|eqtransformer [STEPS]
M=${1:-phasenet}; S=${2:-4000}
python bench.py --model $M --no-cpu-baseline --steps $S --warmup 50 --repeats 4 > gpurun_out/power_$M.json 2>/dev/null &
P=$!
sleep 6   # model load + warm-up
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk|mclk|fclk" | tr '\n' ' ' | sed -E 's/ +/ /g; s/=+//g'
  echo
done
wait $P
python -c "import json; d=json.loads(open('gpurun_out/power_$M.json').read().strip().splitlines()[-1]); print('$M', round(d['value']), d['ms_per_step'])"
