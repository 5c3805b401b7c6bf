#!/bin/bash
# A/B of the HIP kernarg placement (HIP_FORCE_DEV_KERNARG) on both bench lines, plus the mid-kernel phase clock.
mkdir -p gpurun_out
timeout -k 10 100 python tools/mid_clock.py 2>&1 | grep -v amdgpu.ids > gpurun_out/mid_clock.txt
for m in eqtransformer phasenet; do
  for v in default 1 0; do
    if [ $v = default ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
    timeout -k 10 200 python bench.py --model $m --no-cpu-baseline > gpurun_out/ka_${m}_$v.json 2> gpurun_out/ka_${m}_$v.err
  done
done
python - <<PY
import json
for m in ("eqtransformer", "phasenet"):
    for v in ("default", "1", "0"):
        try:
            d = json.loads(open(f"gpurun_out/ka_{m}_{v}.json").read().strip().splitlines()[-1])
            print(m, "HIP_FORCE_DEV_KERNARG=" + v, round(d["value"]), round(d["ms_per_step"], 4), round(d["forward"]["sum_kernel_ms"], 4))
        except Exception as e:
            print(m, v, "ERR", e)
PY
cat gpurun_out/mid_clock.txt
