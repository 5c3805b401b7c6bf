#!/bin/bash
# Who owns the LDS bank conflicts of pn_window_kernel?  The same counter pass (counters only; program directly after --) over
# the plans that are left after round 6's pruning: the default, every core layer on the fp32 MFMA (plan_flags[5] = 3), level 0 on the
# vector ALUs (8; rounds 4-5 also walked the intermediate forms 4 .. 7: profiles/r04_b_, r05_h_, r06_c_lds_bank_conflict_owner.txt), and over the
# three-launch plan whose level-0 kernels (pn_down0v / pn_up3v: the VALU convs of the window kernel, time-tiled) stand alone.
#   usage, on the GPU box:  bash tools/pmc_lds_owner.sh  -> gpurun_out/lds_owner/summary.txt
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/lds_owner; rm -rf $O; mkdir -p $O; cd /tmp
for f in 0 3 8 2; do
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/f$f -- python3 $R/tools/run_forward.py phasenet 4 0,0,0,0,0,$f > $O/f$f.log 2>&1
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass killed (rc=$rc): stopping"; exit $rc; fi
done
cd $R
python3 - "$O" <<'PY' | tee $O/summary.txt
import collections, csv, glob, sys
O = sys.argv[1]
what = {"0": "default: every layer but the strided convs on the bf16 matrix cores", "3": "every core layer on the fp32 MFMA",
        "8": "level 0 in round 4's forms (vector ALUs / fp32 MFMA), the core on the bf16 matrix cores",
        "2": "three launches: level-0 VALU kernels alone, fp32 core alone"}
for f in "0382":
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(f"{O}/f{f}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"]
            if "pn_" in k:
                acc[k.replace("vp::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"plan_flags[5] = {f}: {what[f]}")
    for k, d in sorted(acc.items()):
        m = {c: (sum(v[1:]) / len(v[1:]) if len(v) > 1 else v[0]) / 256.0 for c, v in d.items()}
        print(f"  {k:62s} per window: LDS instructions {m.get('SQ_INSTS_LDS', 0):8.0f}  LDS active cycles {m.get('SQ_LDS_IDX_ACTIVE', 0):8.0f}  "
              f"bank-conflict cycles {m.get('SQ_LDS_BANK_CONFLICT', 0):8.0f}  ratio {m.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, m.get('SQ_LDS_IDX_ACTIVE', 0)):.2f}")
PY
