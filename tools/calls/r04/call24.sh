#!/bin/bash
# round 4, call 24: (d) of tools/hbm_probe -- the same copies over different CONTENT (zeros, constants, hashed words, state-like floats), order A B A
set -u
OUT=gpurun_out/r04_c24; mkdir -p $OUT
export TMPDIR=/tmp
for lg in 22 20 24 21; do timeout 300 tools/hbm_probe $lg data > $OUT/data_probe_2p$lg.log 2>&1; echo "probe 2^$lg rc $?" >> $OUT/status.log; done
echo done >> $OUT/status.log
