#!/bin/bash
# round 4, call 31: CartPole with its constant reward store elided like MountainCar's (tools/patches/r04_exp_elide_cartpole_reward.diff, developer build)
# against the product kernel, by size, both call shapes; plus a bit check of the elided build against the base build (state, flags, rewards, statistics)
set -u
OUT=gpurun_out/r04_c31; mkdir -p $OUT
export TMPDIR=/tmp
for aql in 0 1; do
  for lg in 20 21 22 23 24 25; do
    steps=$(( 6000 >> (lg - 20) ))
    GYMRS_AQL=$aql timeout 600 python tools/step_timer.py --lib _ab/libbase.so --lib _ab/libelidecp.so --n $((1 << lg)) --steps $steps --reps 7 --nbuf 8 > $OUT/elide_2p${lg}_aql$aql.log 2>&1
    echo "2^$lg aql $aql rc $?" >> $OUT/status.log
  done
done
echo done >> $OUT/status.log
