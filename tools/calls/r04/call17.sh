#!/bin/bash
# round 4, call 17: hunting the one segmentation fault of evidence run r04b (at the suite's first subprocess): the first three test files, 40 times, with the
# native-backtrace handler (GYMRS_TEST_SEGV_TRACE=1)
set -u
OUT=gpurun_out/r04_c17; mkdir -p $OUT
export TMPDIR=/tmp GYMRS_TEST_SEGV_TRACE=1
for i in $(seq 1 40); do
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slowpaths.py tests/test_gpu_envs.py -m gpu -x -q -p no:cacheprovider > $OUT/run.log 2>&1; rc=$?
  echo "run $i rc $rc $(grep -E 'passed|failed' $OUT/run.log | tail -1)" >> $OUT/status.log
  if [ $rc -ne 0 ]; then cp $OUT/run.log $OUT/failed_$i.log; fi
done
echo done >> $OUT/status.log
