#!/bin/bash
# Round-3 profile collection on the GPU box (run through gpurun).  Kernel-trace statistics of the bench command and of every
# secondary step, then one PMC pass per counter group and step (a --pmc run is never combined with another trace domain).
# Outputs under gpurun_out/r03_prof/; the summaries to commit land in gpurun_out/r03_prof/summary/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_prof
mkdir -p $OUT/summary
echo "== kernel stats: bench"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o b -- python3 bench.py --steps 100 --warmup 30 --no-cpu-baseline > $OUT/bench.log 2>&1 || exit 1
grep '^{' $OUT/bench.log > $OUT/summary/r03_bench_under_rocprof.json.log
declare -A TARGET=( [train_step]=tools/gpu_prof_train.py [cnn1d_fwd]=tools/gpu_prof_cnn1d.py [cnn1d_train_step]=tools/gpu_prof_cnn1d_train.py \
                    [cae_score]=tools/gpu_prof_cae.py [cae_train_step]=tools/gpu_prof_cae_train.py [fwd]=tools/gpu_prof_fwd.py )
for step in train_step cnn1d_fwd cnn1d_train_step cae_score cae_train_step; do
  echo "== kernel stats: $step"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_$step -o k -- python3 ${TARGET[$step]} > $OUT/ks_$step.log 2>&1 || exit 1
done
SPECS=""
for step in fwd train_step cnn1d_fwd cae_score cae_train_step cnn1d_train_step; do
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $grp | cut -d' ' -f1)
    echo "== pmc $tag: $step"
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_${step}_$tag -o p -- python3 ${TARGET[$step]} > $OUT/pmc_${step}_$tag.log 2>&1 || exit 1
    f=$(find $OUT/pmc_${step}_$tag -name 'p_counter_collection.csv' | head -1)
    SPECS="$SPECS $step:$tag=$f"
  done
done
python3 tools/pmc_traffic_json.py $OUT/summary/r03_pmc_traffic.json $SPECS || exit 1
cp $(find $OUT/bench -name 'b_kernel_stats.csv' | head -1) $OUT/summary/r03_bench_kernel_stats.csv
for step in train_step cnn1d_fwd cnn1d_train_step cae_score cae_train_step; do
  cp $(find $OUT/ks_$step -name 'k_kernel_stats.csv' | head -1) $OUT/summary/r03_${step}_kernel_stats.csv
done
ls -la $OUT/summary
