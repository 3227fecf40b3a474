#!/bin/bash
# Per-kernel stats of the A/B training loop under two builds on the same device (run through gpurun):
#   bash tools/gpu_profile_ab_train.sh libdfa_hip_prev.so libdfa_hip.so
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ab_prof
mkdir -p $OUT
export DFA_AB_MODE=train
for lib in "$@"; do
  export DFA_LIB=$lib
  rm -rf $OUT/$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$lib -o t -- python3 tools/gpu_ab_lib.py > $OUT/$lib.log 2>&1 || exit 1
  grep "train step" $OUT/$lib.log
done
python3 - "$@" <<PY
import csv, glob, sys
tabs = []
for lib in sys.argv[1:]:
    f = glob.glob("$OUT/%s/**/t_kernel_stats.csv" % lib, recursive=True)[0]
    tabs.append({r["Name"]: float(r["TotalDurationNs"]) / 55 / 1e6 for r in csv.DictReader(open(f))})
names = sorted(set().union(*tabs), key=lambda n: -max(t.get(n, 0) for t in tabs))
for n in names:
    v = [t.get(n, 0) for t in tabs]
    if max(v) > 0.02: print("  ".join("%7.4f" % x for x in v), " ", n[:110])
print("  ".join("%7.3f" % sum(t.values()) for t in tabs), "  sum ms/step")
PY
