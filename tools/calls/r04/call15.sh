#!/bin/bash
# round 4, call 15: workgroups of 256 / 128 / 64 work-items (GYMRS_EXP_BLOCK) where a launch is many generations of waves; HIP launches (the chain's code
# object holds the 256 / 512 kernels only); env 0 CartPole, 1 MountainCar, 2 Pendulum
set -u
OUT=gpurun_out/r04_c15; mkdir -p $OUT
export TMPDIR=/tmp
for env in 0 2 1; do
  for lg in 22 23 24 25; do
    [ $env = 2 ] && [ $lg = 25 ] && continue
    GYMRS_AQL=0 timeout 600 python tools/step_timer.py --env $env --n $((1 << lg)) --lib _ab/libb256.so --lib _ab/libb128.so --lib _ab/libb64.so --steps $((6000 >> (lg - 20))) --reps 5 --nbuf 8 > $OUT/block_env${env}_2p${lg}.log 2>&1
  done
done
echo done >> $OUT/status.log
