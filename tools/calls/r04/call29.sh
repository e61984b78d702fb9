#!/bin/bash
# round 4, call 29: long-run check of the chain path against HIP launches (tests/soak_chains.py): 2e6 steps of 2^20 lanes per env
set -u
OUT=gpurun_out/r04_c29; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python tests/soak_chains.py 2000000 > $OUT/soak_chains.log 2>&1; echo "soak rc $?" >> $OUT/status.log
echo done >> $OUT/status.log
