# kernel-trace statistics of the CNN2D training step (tools/gpu_prof_train.py): per-kernel average durations, top 16
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/train_ks
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o k -- python3 tools/gpu_prof_train.py > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/train_ks/ks/**/k_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:16]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us x{r['Calls']:>4} {float(r['Percentage']):5.1f}%  {r['Name'][:110]}")
PY
