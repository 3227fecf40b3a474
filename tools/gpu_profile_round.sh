#!/bin/bash
# Round profile collection on the GPU box (run through gpurun): kernel stats of the bench command and of the training step,
# then one PMC pass per counter group (never combined with other trace domains).  Outputs under gpurun_out/r02_prof/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02_prof
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o b -- python3 bench.py --steps 100 --warmup 30 --no-cpu-baseline > $OUT/bench.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -o t -- python3 tools/gpu_prof_train.py > $OUT/train.log 2>&1 || exit 1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_fwd_$tag -o p -- python3 tools/gpu_prof_fwd.py > $OUT/pmc_fwd_$tag.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_train_$tag -o p -- python3 tools/gpu_prof_train.py > $OUT/pmc_train_$tag.log 2>&1 || exit 1
done
ls -R $OUT | head -60
