#!/bin/bash
# SQ issue / wait breakdown of the forward kernels (one PMC pass per counter group; kernel trace only).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02_sq
mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -o p -- python3 tools/gpu_prof_fwd.py > $OUT/g$i.log 2>&1 || { tail -5 $OUT/g$i.log; exit 1; }
done
python3 - <<PY
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/g*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv12_fused" in k or "conv3_m16" in k or "conv_split" in k or "conv1_bn" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/sq_summary.csv", "w") as out:
    w = csv.writer(out)
    names = sorted({c for k in acc for c in acc[k]})
    w.writerow(["Kernel"] + names)
    for k in sorted(acc):
        w.writerow([k] + [round(sum(acc[k][c]) / len(acc[k][c]), 1) if c in acc[k] else "" for c in names])
print(open("$OUT/sq_summary.csv").read())
PY
