#!/bin/bash
# SQ issue / wait / LDS breakdown of the training-step kernels (one PMC pass per counter group; kernel trace only).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02_sq_train
mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -o p -- python3 tools/gpu_prof_train.py > $OUT/g$i.log 2>&1 || { tail -5 $OUT/g$i.log; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, collections, glob, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(s in k for s in ("wgrad3x3_bf16", "conv_split", "conv3_m16", "conv3x3_mfma", "conv1_mfma", "bn_")):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/sq_train_summary.csv", "w") as o:
    w = csv.writer(o)
    names = sorted({c for k in acc for c in acc[k]})
    w.writerow(["Kernel"] + names)
    for k in sorted(acc):
        w.writerow([k] + [round(sum(acc[k][c]) / len(acc[k][c]), 1) if c in acc[k] else "" for c in names])
for k in sorted(acc):
    a = {c: sum(v) / len(v) for c, v in acc[k].items()}
    wc = a.get("SQ_WAVE_CYCLES", 0) or 1
    print("%-70s" % k[:70])
    print("    wait_any %.2f  wait_inst_any %.2f  wait_lds %.2f  active_any %.2f | valu/mfma %.2f  lds/mfma %.2f | lds_conflict/lds_active %.2f | mfma_busy_cycles %.3g" % (
        a.get("SQ_WAIT_ANY", 0) / wc, a.get("SQ_WAIT_INST_ANY", 0) / wc, a.get("SQ_WAIT_INST_LDS", 0) / wc, a.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        (a.get("SQ_INSTS_VALU", 0) - a.get("SQ_INSTS_MFMA", 0)) / max(a.get("SQ_INSTS_MFMA", 0), 1), a.get("SQ_INSTS_LDS", 0) / max(a.get("SQ_INSTS_MFMA", 0), 1),
        a.get("SQ_LDS_BANK_CONFLICT", 0) / max(a.get("SQ_LDS_IDX_ACTIVE", 0), 1), a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)))
PY
