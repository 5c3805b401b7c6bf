#!/bin/bash
# A/B of two builds of the library on ONE box: bench.py runs interleaved, A B A B ...
# usage: tools/ab_lib.sh ROUNDS LIB_A LIB_B MODEL [bench args]     (LIB = path of a libvolpick_hip.so)
R=$1; A=$2; B=$3; M=$4; shift 4
mkdir -p gpurun_out/ab
for i in $(seq 1 $R); do
  for v in A B; do
    if [ $v = A ]; then f="$A"; else f="$B"; fi
    VOLPICK_HIP_LIB="$PWD/$f" timeout -k 10 200 python bench.py --model $M --no-cpu-baseline --steps 50 "$@" > gpurun_out/ab/lib${v}_$i.json 2> gpurun_out/ab/lib${v}_$i.err
  done
done
python - "$R" "$A" "$B" <<'PY'
import json, sys, statistics
R = int(sys.argv[1])
for v, f in zip("AB", sys.argv[2:4]):
    vals = []
    for i in range(1, R + 1):
        try:
            d = json.loads(open(f"gpurun_out/ab/lib{v}_{i}.json").read().strip().splitlines()[-1])
            vals.append((d["value"], d["forward"]["sum_kernel_ms"]))
        except Exception as e:
            print(v, i, "ERR", e)
    print(v, f, "windows/s", [round(x) for x, _ in vals], "median", round(statistics.median(x for x, _ in vals)),
          " sum of launches (us)", [round(s * 1e3, 1) for _, s in vals])
PY
