#!/bin/bash
# SQ counter passes over the forward kernels of BOTH models (counters only: --kernel-trace + --pmc, program directly after --;
# one small counter group per pass).   usage, on the GPU box:  bash tools/pmc_sq_all.sh TAG [iterations] ["models"]
# -> gpurun_out/sq_TAG/{phasenet,eqtransformer}_gN/ (raw CSVs) and gpurun_out/sq_TAG/summary.txt: per kernel the mean of
#    every counter over the launches after the first, and what follows from them:
#      mfma_busy      = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES: share of the kernel's busy time in which a matrix
#                       pipe works (SQ_VALU_MFMA_BUSY_CYCLES counts per SIMD-cycle summed over the chip and is
#                       normalised by 4 SIMDs x the CU-cycles of SQ_BUSY_CU_CYCLES)
#      valu_per_mfma  = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA: vector instructions issued per matrix instruction
export TMPDIR=/tmp
T=${1:-a}; N=${2:-4}; R=$PWD; O=$R/gpurun_out/sq_$T; rm -rf $O; mkdir -p $O; cd /tmp
GROUPS_=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32"
 "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"
 "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
 "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE"
)
for model in ${3:-phasenet eqtransformer}; do
  i=0
  for grp in "${GROUPS_[@]}"; do
    i=$((i+1))
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/${model}_g$i -- python3 $R/tools/run_forward.py $model $N > $O/${model}_g$i.log 2>&1
    rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass killed (rc=$rc): stopping"; exit $rc; fi
    echo "$model group $i rc=$rc"
  done
done
cd $R
python3 - "$O" <<'PY' | tee $O/summary.txt
import collections, csv, glob, sys
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/*_g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(s in k for s in ("pn_window", "eqt_", "gather_normalize", "stack_kernel", "trigger_scan")):
            k = k.replace("vp::(anonymous namespace)::", "").replace("vp::", "").replace("void ", "")
            acc[k.split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    m = {c: (sum(v[1:]) / len(v[1:]) if len(v) > 1 else v[0]) for c, v in d.items()}
    print(k)
    for c in sorted(m):
        print(f"    {c:34s} {m[c]:18.0f}")
    g = m.get
    if g("SQ_BUSY_CU_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
        print(f"    {'mfma_busy (of 4 SIMD x CU-busy)':34s} {g('SQ_VALU_MFMA_BUSY_CYCLES') / (4.0 * g('SQ_BUSY_CU_CYCLES')):18.3f}")
    if g("SQ_INSTS_MFMA"):
        print(f"    {'valu_per_mfma':34s} {(g('SQ_INSTS_VALU', 0) - g('SQ_INSTS_MFMA')) / g('SQ_INSTS_MFMA'):18.2f}")
    if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_INST_ANY") is not None:
        w = g("SQ_WAVE_CYCLES")
        print(f"    {'wave time: active/wait_inst/wait':34s} {g('SQ_ACTIVE_INST_ANY', 0) / w:6.2f} {g('SQ_WAIT_INST_ANY', 0) / w:6.2f} {g('SQ_WAIT_ANY', 0) / w:6.2f}")
PY
