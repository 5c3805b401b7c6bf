# SQ counters of the training step's forward convs (one small group per pass; counters only): bash tools/pmc_sq_train.sh
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/pmc_sq_train; rm -rf $O; mkdir -p $O; cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CU_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g$i -- python3 $R/tools/bench_train.py --batch 512 --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
cd $R
python - <<'PY' | tee gpurun_out/pmc_sq_train/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_sq_train/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_mfma_kernel" in k:
            acc[k.split("ConvCfg")[1][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    g = {c: sum(v) / len(v) for c, v in d.items()}
    busy = g.get("SQ_BUSY_CU_CYCLES", 0)
    print(k)
    print("   ", "  ".join(f"{c[3:]}={v:.3g}" for c, v in sorted(g.items())))
    if busy:
        print(f"    mfma_busy/(4*CU-busy) = {g.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * busy):.3f}   lds_conflict/active = {g.get('SQ_LDS_BANK_CONFLICT', 0) / max(g.get('SQ_LDS_IDX_ACTIVE', 1), 1):.3f}   wave: active {g.get('SQ_ACTIVE_INST_ANY', 0) / max(g.get('SQ_WAVE_CYCLES', 1), 1):.2f} wait_inst {g.get('SQ_WAIT_INST_ANY', 0) / max(g.get('SQ_WAVE_CYCLES', 1), 1):.2f} wait {g.get('SQ_WAIT_ANY', 0) / max(g.get('SQ_WAVE_CYCLES', 1), 1):.2f}")
PY
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
