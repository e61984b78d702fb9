#!/bin/bash
# round 4, call 8: where did MountainCar's 0.3 us go?  The library at this round's commits, alternately (HIP launches and chains)
set -u
OUT=gpurun_out/r04_c8; mkdir -p $OUT
export TMPDIR=/tmp
for env in 1 0; do
  GYMRS_AQL=0 timeout 600 python tools/step_timer.py --env $env --lib _ab/libr03.so --lib _ab/lib_f4b3604.so --lib _ab/lib_9fe2185.so --lib gym-rs_amd/libgymrs_amd.so --lib _ab/libnoxcc.so --steps 5000 --reps 9 > $OUT/bisect_env${env}_hip.log 2>&1
  GYMRS_AQL=1 timeout 600 python tools/step_timer.py --env $env --lib _ab/libr03.so --lib _ab/lib_9fe2185.so --lib gym-rs_amd/libgymrs_amd.so --lib _ab/libnoxcc.so --steps 5000 --reps 9 > $OUT/bisect_env${env}_chain.log 2>&1
done
# the same pairs in the other order (is it the position in the process?)
GYMRS_AQL=0 timeout 600 python tools/step_timer.py --env 1 --lib gym-rs_amd/libgymrs_amd.so --lib _ab/libr03.so --steps 5000 --reps 9 > $OUT/order_env1_hip.log 2>&1
GYMRS_AQL=0 timeout 600 python tools/step_timer.py --env 1 --lib gym-rs_amd/libgymrs_amd.so --steps 5000 --reps 9 > $OUT/alone_now_env1_hip.log 2>&1
GYMRS_AQL=0 timeout 600 python tools/step_timer.py --env 1 --lib _ab/libr03.so --steps 5000 --reps 9 > $OUT/alone_r03_env1_hip.log 2>&1
echo done >> $OUT/status.log
