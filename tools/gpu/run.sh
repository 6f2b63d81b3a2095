#!/bin/bash
# One parameterised GPU call script (replaces the per-call g*.sh files of rounds 1-4):
#   gpurun --timeout N -- 'bash tools/gpu/run.sh <name> "<command>" ["<command>" ...]'
# Every command's output goes to gpurun_out/<name>_<i>.log; the exit code is that of the last failing command (0 if all pass).
name=$1; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rc=0; i=0
for c in "$@"; do
  i=$((i + 1))
  echo "=== [$name $i] $c" | tee -a gpurun_out/${name}.log
  bash -c "$c" > gpurun_out/${name}_$i.log 2>&1 || rc=$?
  tail -n 25 gpurun_out/${name}_$i.log
done
exit $rc
