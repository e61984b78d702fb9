#!/bin/bash
# round 4, call 12: the wave's variants VERDICT r3 "next" #7 names: outputs stored before the re-arm pass; 4 instead of 8 waves per workgroup
set -u
OUT=gpurun_out/r04_c12; mkdir -p $OUT
export TMPDIR=/tmp
for env in 0 2; do
  for aql in 1 0; do
    GYMRS_AQL=$aql timeout 600 python tools/step_timer.py --env $env --lib _ab/libbase.so --lib _ab/libearly.so --steps 5000 --reps 9 > $OUT/early_env${env}_aql$aql.log 2>&1
  done
done
GYMRS_AQL=1 timeout 600 python tools/step_timer.py --env 0 --lib _ab/libbase.so --steps 5000 --reps 9 > $OUT/threads512_env0_aql1.log 2>&1
GYMRS_AQL=1 GYMRS_DEV_THREADS=256 timeout 600 python tools/step_timer.py --env 0 --lib _ab/libbase.so --steps 5000 --reps 9 > $OUT/threads256_env0_aql1.log 2>&1
for lg in 21 22 23; do
  GYMRS_AQL=1 timeout 600 python tools/step_timer.py --env 0 --n $((1<<lg)) --lib _ab/libbase.so --steps $((6000 >> (lg-20))) --reps 5 --nbuf 8 > $OUT/threads512_2p${lg}.log 2>&1
  GYMRS_AQL=1 GYMRS_DEV_THREADS=256 timeout 600 python tools/step_timer.py --env 0 --n $((1<<lg)) --lib _ab/libbase.so --steps $((6000 >> (lg-20))) --reps 5 --nbuf 8 > $OUT/threads256_2p${lg}.log 2>&1
done
echo done >> $OUT/status.log
