#!/bin/bash
# round 4, call 9: which part of the XCD check costs MountainCar's chain 0.3 us?  GYMRS_EXP_XCC_PARTS: 1 table fetch, 2 s_getreg, 4 compare / record at the end
set -u
OUT=gpurun_out/r04_c9; mkdir -p $OUT
export TMPDIR=/tmp
for env in 1 0; do
  GYMRS_AQL=1 timeout 600 python tools/step_timer.py --env $env --lib _ab/libr03.so --lib _ab/libx0.so --lib _ab/libx1.so --lib _ab/libx2.so --lib _ab/libx4.so --lib _ab/libx3.so --lib _ab/libx6.so --lib _ab/libx7.so --steps 5000 --reps 7 > $OUT/parts_env${env}_chain.log 2>&1
done
echo done >> $OUT/status.log
