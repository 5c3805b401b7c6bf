#!/bin/bash
# End-to-end A/B of EQTransformer plan variants on ONE box: bench.py runs interleaved, A B A B ... (box-to-box spread is
# larger than most plan effects).  usage: tools/ab_e2e.sh ROUNDS FLAGS_A FLAGS_B [bench args]
R=$1; A=$2; B=$3; shift 3
mkdir -p gpurun_out/ab
for i in $(seq 1 $R); do
  for v in A B; do
    if [ $v = A ]; then f="$A"; else f="$B"; fi
    VOLPICK_PLAN_FLAGS="$f" timeout -k 10 200 python bench.py --model eqtransformer --no-cpu-baseline --steps 50 "$@" > gpurun_out/ab/${v}_$i.json 2> gpurun_out/ab/${v}_$i.err
  done
done
python - "$R" "$A" "$B" <<'PY'
import json, sys, statistics
R = int(sys.argv[1])
for v, f in zip("AB", sys.argv[2:4]):
    vals = []
    for i in range(1, R + 1):
        try:
            d = json.loads(open(f"gpurun_out/ab/{v}_{i}.json").read().strip().splitlines()[-1])
            vals.append((d["value"], d["forward"]["sum_kernel_ms"]))
        except Exception as e:
            print(v, i, "ERR", e)
    print(v, f"flags={f!r}", "windows/s", [round(x) for x, _ in vals], "median", round(statistics.median(x for x, _ in vals)),
          " sum of launches (us)", [round(s * 1e3, 1) for _, s in vals])
PY
