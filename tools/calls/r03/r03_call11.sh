#!/bin/bash
set -u
OUT=gpurun_out/r03_c11
mkdir -p $OUT
for v in "GYMRS_AQL=0" "X=1" "GYMRS_AQL_LAZY_WAIT=1" "GYMRS_AQL_LAZY_WAIT=1 GYMRS_AQL_FENCES=11"; do
  echo "== engine $v, wall clock, 16 chains of 2000 steps" >> $OUT/engine.log
  env $v timeout 100 python tools/step_timer.py --env 0 --n 1048576 --steps 2000 --reps 16 --all 1 --wall 1 2>&1 | grep -v "amdgpu.ids\|ring at" >> $OUT/engine.log
done
cat $OUT/engine.log
