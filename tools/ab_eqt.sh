#!/bin/bash
# A/B of two EQTransformer plans on one box: bench lines + per-launch times.  usage: ab_eqt.sh FLAGS_A FLAGS_B
mkdir -p gpurun_out
for v in A B; do
  if [ $v = A ]; then f="$1"; else f="$2"; fi
  VOLPICK_PLAN_FLAGS="$f" timeout -k 10 200 python bench.py --model eqtransformer --no-cpu-baseline --sustain-seconds 0 --no-api > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
done
python - "$1" "$2" <<'PY'
import json, sys
d = {}
for v, f in zip("AB", sys.argv[1:]):
    try:
        d[v] = json.loads(open(f"gpurun_out/ab_{v}.json").read().strip().splitlines()[-1])
        print(v, f"flags={f!r}", round(d[v]["value"]), "windows/s", round(d[v]["ms_per_step"], 4), "ms/step  sum of launches", round(d[v]["forward"]["sum_kernel_ms"], 4))
    except Exception as e:
        print(v, "ERR", e, open(f"gpurun_out/ab_{v}.err").read()[-400:])
if len(d) == 2:
    for ka, kb in zip(d["A"]["forward"]["kernels"], d["B"]["forward"]["kernels"]):
        print(f"  {ka['name'][:34]:34s} {ka['ms']*1e3:8.1f} {kb['ms']*1e3:8.1f} us")
PY
