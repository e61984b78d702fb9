#!/bin/bash
# round 4, call 20: the GPU suite on every submission path (as round 3's profiles/r03_alt_paths_suite.log): HIP launches only, synchronous hand-over
# everywhere, calibration forced to the synchronous hand-over; the perf-marked tests are left out (they assert the default path's rates)
set -u
OUT=gpurun_out/r04_c20; mkdir -p $OUT
export TMPDIR=/tmp
GYMRS_AQL=0 timeout 900 python -m pytest tests -m "gpu and not perf" -x -q -p no:cacheprovider > $OUT/suite_aql0.log 2>&1; echo "GYMRS_AQL=0 rc $? $(grep -E 'passed|failed' $OUT/suite_aql0.log | tail -1)" >> $OUT/status.log
GYMRS_AQL_SYNC=1 timeout 900 python -m pytest tests -m "gpu and not perf" -x -q -p no:cacheprovider > $OUT/suite_sync.log 2>&1; echo "GYMRS_AQL_SYNC=1 rc $? $(grep -E 'passed|failed' $OUT/suite_sync.log | tail -1)" >> $OUT/status.log
GYMRS_AQL_HANDOVER=sync timeout 900 python -m pytest tests -m "gpu and not perf" -x -q -p no:cacheprovider > $OUT/suite_handover_sync.log 2>&1; echo "GYMRS_AQL_HANDOVER=sync rc $? $(grep -E 'passed|failed' $OUT/suite_handover_sync.log | tail -1)" >> $OUT/status.log
echo done >> $OUT/status.log
