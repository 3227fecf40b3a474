# LDS bank-conflict counters of the CNN2D eval forward's kernels (one PMC pass, kernel trace only): conflict cycles / LDS-active cycles
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/lds_conf
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p -o p -- python3 tools/gpu_prof_fwd.py > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/lds_conf/p/**/p_counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
with open('gpurun_out/lds_conf/summary.csv', 'w') as o:
    o.write('kernel,launches,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,ratio\n')
    for k, v in acc.items():
        c = v.get('SQ_LDS_BANK_CONFLICT', [0]); a = v.get('SQ_LDS_IDX_ACTIVE', [0])
        mc, ma = sum(c) / len(c), sum(a) / len(a)
        line = f'"{k}",{len(c)},{mc:.0f},{ma:.0f},{(mc / ma if ma else 0):.4f}'
        o.write(line + '\n'); print(line)
PY
