# kernel-trace statistics of one profile target: tools/gpu_ks.sh <target.py> [top N]  -> per-kernel average / total per step
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=$1; N=${2:-24}
OUT=gpurun_out/ks_$(basename $T .py)
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o k -- python3 $T > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 - $OUT $N <<'PY'
import csv, glob, sys
out, n = sys.argv[1], int(sys.argv[2])
f = glob.glob(out + '/ks/**/k_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6:.3f} ms over the run")
for r in rows[:n]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us x{r['Calls']:>4} {float(r['Percentage']):5.1f}%  {r['Name'][:120]}")
PY
