#!/bin/bash
set -u
OUT=gpurun_out/r03_c10
mkdir -p $OUT
for v in "X=1" "GYMRS_AQL_QUEUE=16384" "GYMRS_AQL_QUEUE=1024" "GYMRS_AQL_FLUSH=1"; do
  echo "== engine $v, 40 chains of 500 steps" >> $OUT/engine.log
  env $v timeout 100 python tools/step_timer.py --env 0 --n 1048576 --steps 500 --reps 40 --all 1 2>&1 | grep -v "amdgpu.ids\|ring at" >> $OUT/engine.log
done
cat $OUT/engine.log
