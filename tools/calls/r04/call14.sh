#!/bin/bash
# round 4, call 14: the mid-size question in the probe: the step's nine streams WITH arithmetic between loads and stores (FMA rounds), LDS padding, tiles
set -u
OUT=gpurun_out/r04_c14; mkdir -p $OUT
for lg in 21 22 23 24; do timeout 400 tools/hbm_probe $lg 2>&1 | sed -n '/^# (b)/,$p' | grep -v "skew      0\|skew   9472\|skew 261376\|skew  12544\|out of place" > $OUT/hbm_probe_alu_2p$lg.log; done
echo done >> $OUT/status.log
