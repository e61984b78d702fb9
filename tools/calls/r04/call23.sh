#!/bin/bash
# round 4, call 23: (c)(iv) of tools/hbm_probe -- one generation of waves per launch: the streams of a step as k launches of n / k lanes
set -u
OUT=gpurun_out/r04_c23; mkdir -p $OUT
export TMPDIR=/tmp
for lg in 20 21 22 23 24; do timeout 300 tools/hbm_probe $lg phase > $OUT/phase_probe_2p$lg.log 2>&1; echo "probe 2^$lg rc $?" >> $OUT/status.log; done
echo done >> $OUT/status.log
