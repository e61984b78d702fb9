#!/bin/bash
# round 4, call 22: the GPU suite with the tests' child processes started by tests/spawn_server.py; (c) of tools/hbm_probe again with the loads-only /
# stores-only modes; chip-wide per-wave timelines (tools/trace, time bases clustered) at 2^20 / 2^22 / 2^23 lanes, plain and with the first generation staggered
set -u
OUT=gpurun_out/r04_c22; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $? $(grep -E 'passed|failed' $OUT/pytest_gpu.log | tail -1)" >> $OUT/status.log
for lg in 22 23; do timeout 300 tools/hbm_probe $lg phase > $OUT/phase_probe_2p$lg.log 2>&1; echo "probe 2^$lg rc $?" >> $OUT/status.log; done
for lg in 20 22 23; do
  GYMRS_AQL=0 timeout 120 tools/trace 3 13 $lg 1200 > $OUT/timeline_2p${lg}_plain.log 2>&1; echo "timeline rc $?" >> $OUT/status.log
  GYMRS_AQL=0 LD_LIBRARY_PATH=_ab/stag4 timeout 120 tools/trace 3 13 $lg 1200 > $OUT/timeline_2p${lg}_stag4.log 2>&1; echo "timeline stag rc $?" >> $OUT/status.log
done
echo done >> $OUT/status.log
