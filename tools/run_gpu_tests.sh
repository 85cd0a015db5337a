#!/bin/bash
# GPU box: every GPU test file in its own process (a GPU fault aborts the interpreter: the other files still run).
# usage: tools/run_gpu_tests.sh <outdir> [pytest args]
OUT=${1:-gpurun_out/tests}; shift || true
mkdir -p $OUT
rc=0
for f in tests/test_gpu_*.py; do
  n=$(basename $f .py)
  timeout 900 python -m pytest $f -m gpu -q -x "$@" > $OUT/$n.log 2>&1
  c=$?
  echo "$n exit $c: $(grep -E 'passed|failed|error' $OUT/$n.log | tail -1)"
  [ $c -ne 0 ] && rc=1
done
exit $rc
