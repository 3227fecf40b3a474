#!/bin/bash
# rocprofv3 kernel stats of one profile target (run through gpurun): gpu_profile_one.sh <tag> <script.py> <passes>
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02_prof
mkdir -p $OUT
rm -rf $OUT/$1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$1 -o k -- python3 $2 > $OUT/$1.log 2>&1 || { tail -5 $OUT/$1.log; exit 1; }
python3 - "$OUT/$1" "$3" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/k_kernel_stats.csv", recursive=True)[0]
n = float(sys.argv[2])
tot = 0
for r in csv.DictReader(open(f)):
    ms = float(r["TotalDurationNs"]) / n / 1e6
    tot += ms
    if ms > 0.01: print("%7.4f ms/pass  calls %4.1f  %s" % (ms, int(r["Calls"]) / n, r["Name"][:110]))
print("sum %.3f ms/pass" % tot)
PY
