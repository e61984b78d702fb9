#!/bin/bash
# round 4, call 25: the engine's array skew (GYMRS_ARRAY_SKEW: array k of the pool starts k * skew bytes past its power-of-two place) on the REAL CartPole
# kernel, both call shapes, 2^20 .. 2^22 lanes: the library's single-stream copy of non-zero bytes is 5-14 % faster than a nine-stream copy of the same bytes
set -u
OUT=gpurun_out/r04_c25; mkdir -p $OUT
export TMPDIR=/tmp
for aql in 1 0; do
  for lg in 20 21 22 23; do
    steps=$(( 6000 >> (lg - 20) ))
    for skew in 4352 0 256 1024 2304 8448 16640 33024 65792 131328 262400 1048832 4352; do
      r=$(GYMRS_AQL=$aql GYMRS_ARRAY_SKEW=$skew timeout 300 python tools/step_timer.py --n $((1 << lg)) --steps $steps --reps 5 --nbuf 8 2>&1 | grep median | head -1)
      echo "GYMRS_AQL=$aql 2^$lg lanes skew $skew: $r" >> $OUT/skew_cartpole.log
    done
  done
done
echo done >> $OUT/status.log
