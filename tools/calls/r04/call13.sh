#!/bin/bash
# round 4, call 13: a ring of action buffers that does NOT fit the Infinity Cache (32 buffers at 2^22 Pendulum lanes = 512 MiB): do hinted action loads protect the state's residency?
# h8 = outputs streamed (the product's rule at this size), h12 = outputs + action loads streamed
set -u
OUT=gpurun_out/r04_c13; mkdir -p $OUT
export TMPDIR=/tmp
for aql in 1 0; do
  for nbuf in 32 8; do
    GYMRS_AQL=$aql timeout 600 python tools/step_timer.py --env 2 --n 4194304 --lib _ab/libh8.so --lib _ab/libh12.so --steps 1500 --reps 5 --nbuf $nbuf > $OUT/pendulum_2p22_nbuf${nbuf}_aql$aql.log 2>&1
    GYMRS_AQL=$aql timeout 600 python tools/step_timer.py --env 0 --n 16777216 --lib _ab/libh8.so --lib _ab/libh12.so --steps 400 --reps 5 --nbuf $nbuf > $OUT/cartpole_2p24_nbuf${nbuf}_aql$aql.log 2>&1
  done
done
echo done >> $OUT/status.log
