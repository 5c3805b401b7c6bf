# SQ counters of the PhaseNet forward kernels (one small group per pass; counters only).
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/pmc_sq; rm -rf $O; mkdir -p $O; cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g$i -- python3 $R/tools/run_forward.py ${1:-phasenet} 4 > /dev/null 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_sq/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "pn_" in k or "conv_mfma" in k or "eqt_res" in k:
            acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        v = v[1:] or v
        print(f"    {c:32s} {sum(v)/len(v):16.0f}")
PY
