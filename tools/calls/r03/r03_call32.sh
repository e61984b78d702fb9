#!/bin/bash
set -u
OUT=gpurun_out/r03_c32
mkdir -p $OUT
for cfg in "0 1048576" "1 1048576" "2 4194304" "0 16384" "0 131072"; do
  set -- $cfg
  for k in 8 16 32 64 128; do
    for a in 1 0; do
      echo "== env $1 n $2 steps-per-call $k GYMRS_AQL=$a" >> $OUT/breakeven.log
      GYMRS_AQL=$a timeout 100 python tools/step_timer.py --env $1 --n $2 --steps $k --reps 40 --wall 1 2>&1 | grep -v "amdgpu.ids\|ring at" >> $OUT/breakeven.log
    done
  done
done
cat $OUT/breakeven.log
