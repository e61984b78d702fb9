#!/bin/bash
# round 4, call 16: the GPU suite died once with a segmentation fault while a test was starting a subprocess (evidence run r04b, 1 of 5 full runs that
# day).  Five runs of the suite with a native-backtrace handler installed (GYMRS_TEST_SEGV_TRACE=1, tests/cpp/segv_trace.c)
set -u
OUT=gpurun_out/r04_c16; mkdir -p $OUT
export TMPDIR=/tmp GYMRS_TEST_SEGV_TRACE=1
for i in 1 2 3 4 5; do
  timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/suite_$i.log 2>&1; echo "run $i rc $?" >> $OUT/status.log
  tail -3 $OUT/suite_$i.log | head -2 >> $OUT/status.log
done
echo done >> $OUT/status.log
