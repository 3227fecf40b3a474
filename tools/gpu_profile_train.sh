#!/bin/bash
# Kernel stats of the bf16 training step only (run through gpurun) -> gpurun_out/r02_prof/train, summary printed per step.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02_prof
mkdir -p $OUT
rm -rf $OUT/train
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -o t -- python3 tools/gpu_prof_train.py > $OUT/train.log 2>&1 || exit 1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/train/**/t_kernel_stats.csv", recursive=True)[0]
tot = 0
for r in csv.DictReader(open(f)):
    ms = float(r["TotalDurationNs"]) / 6 / 1e6
    tot += ms
    if ms > 0.02: print(f'{ms:7.4f} ms/step  {r["Name"][:100]}')
print("sum %.3f ms/step" % tot)
PY
