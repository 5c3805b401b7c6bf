# on-box: gpurun -- 'bash tools/train_profile.sh [fp32|bf16]'  -> gpurun_out/train_prof_<dtype>/kernel_stats.csv
export TMPDIR=/tmp
D=${1:-fp32}
python tools/bench_train.py --batch 512 --dtype $D 2>&1 | tail -1
python tools/bench_train.py --batch 1024 --dtype $D --no-cpu-baseline 2>&1 | tail -1
R=$PWD; O=$R/gpurun_out/train_prof_$D; rm -rf $O; mkdir -p $O; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/bench_train.py --batch 512 --dtype $D --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $R; find $O -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time per step: {tot / 13 / 1e3:.0f} us over {sum(int(r['Calls']) for r in rows) // 13} launches")
for r in rows[:45]:
    print(f"{float(r['TotalDurationNs']) / 13 / 1e3:8.1f} us/step {int(r['Calls']) // 13:4d} calls/step {float(r['AverageNs']) / 1e3:8.1f} us avg  {r['Name'][:150]}")
PY
