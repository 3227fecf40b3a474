#!/bin/bash
# Run a list of GPU steps in order on the GPU box (through gpurun), each under its own timeout, logging to gpurun_out/<tag>/.
# A step that fails its assertions does not stop the list; a step that is KILLED or TIMES OUT (rc 124 / 137 / 143) does:
# nothing further touches the GPU after that.   usage: tools/gpu_r3_step.sh <tag> "<name>|<timeout s>|<command>" ...
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; tmo=${rest%%|*}; cmd=${rest#*|}
  echo "=== $name (timeout ${tmo}s): $cmd"
  timeout -k 10 $tmo bash -c "$cmd" > $OUT/$name.log 2>&1
  rc=$?
  echo "=== $name rc=$rc"; tail -n 6 $OUT/$name.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 143 ]; then echo "step $name was killed: stopping"; exit $rc; fi
done
exit 0
