#!/bin/bash
# round 4, call 7: what the XCD check costs with its reads behind every load of the step (all three envs; with the check switched off for comparison)
set -u
OUT=gpurun_out/r04_c7; mkdir -p $OUT
export TMPDIR=/tmp
for env in 0 1 2; do
  GYMRS_AQL=1 timeout 300 python tools/step_timer.py --env $env --lib _ab/libr03.so --lib gym-rs_amd/libgymrs_amd.so --steps 5000 --reps 9 > $OUT/ab_env${env}_check.log 2>&1
  GYMRS_AQL=1 GYMRS_DEV_NO_XCC_CHECK=1 timeout 300 python tools/step_timer.py --env $env --lib _ab/libr03.so --lib gym-rs_amd/libgymrs_amd.so --steps 5000 --reps 9 > $OUT/ab_env${env}_nocheck.log 2>&1
  GYMRS_AQL=0 timeout 300 python tools/step_timer.py --env $env --lib _ab/libr03.so --lib gym-rs_amd/libgymrs_amd.so --steps 5000 --reps 9 > $OUT/ab_env${env}_hip.log 2>&1
done
GYMRS_AQL=1 timeout 300 python tools/step_timer.py --env 2 --n 4194304 --lib _ab/libr03.so --lib gym-rs_amd/libgymrs_amd.so --steps 1500 --reps 7 > $OUT/ab_env2_2p22_check.log 2>&1
echo done >> $OUT/status.log
