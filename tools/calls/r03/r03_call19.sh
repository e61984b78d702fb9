#!/bin/bash
set -u
OUT=gpurun_out/r03_c19
mkdir -p $OUT
for n in 1048576 2097152 4194304 8388608 16777216; do
  echo "== cartpole n=$n chains: wbase (78 VGPRs) / w7 (72, spills) / w8 (64, spills)" >> $OUT/occ.log
  timeout 200 python tools/step_timer.py --lib _ab/libwbase.so --lib _ab/libw7.so --lib _ab/libw8.so --env 0 --n $n --steps 1000 --reps 5 2>&1 | grep -v "amdgpu.ids\|ring at" >> $OUT/occ.log
done
for v in "X=1" "GYMRS_DEV_THREADS=256"; do
  echo "== main lib cartpole 2^20 chains $v" >> $OUT/occ.log
  env $v timeout 100 python tools/step_timer.py --env 0 --n 1048576 --steps 2000 --reps 5 2>&1 | grep -v "amdgpu.ids\|ring at" >> $OUT/occ.log
done
cat $OUT/occ.log
