#!/bin/bash
# kernel statistics of the auto-encoder training step (run through gpurun): gpurun_out/cae_ks/k_kernel_stats.csv
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/cae_ks
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python3 tools/gpu_prof_cae_train.py > $OUT/run.log 2>&1 || exit 1
f=$(find $OUT -name 'k_kernel_stats.csv' | head -1)
cp $f $OUT/stats.csv
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/cae_ks/stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('sum per step ms', tot/6/1e6)
for r in rows[:40]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls'])//6:>3d} {float(r['AverageNs'])/1e3:8.1f} {float(r['TotalDurationNs'])/6e3:8.1f}")
PY
