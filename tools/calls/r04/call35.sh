#!/bin/bash
# round 4, call 35: (e) of tools/hbm_probe LG data -- the state as ONE float4-per-lane array instead of four columns (6 streams instead of 9), non-zero bytes
set -u
OUT=gpurun_out/r04_c35; mkdir -p $OUT
export TMPDIR=/tmp
for lg in 20 21 22 24; do timeout 300 tools/hbm_probe $lg data > $OUT/data_probe_2p$lg.log 2>&1; echo "probe 2^$lg rc $?" >> $OUT/status.log; done
echo done >> $OUT/status.log
