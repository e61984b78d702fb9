#!/bin/bash
set -u
OUT=gpurun_out/r03_c29
mkdir -p $OUT
for n in 1048576 2097152 4194304 8388608 16777216 33554432; do
  for lib in vbase vnolog; do
    echo "== cartpole n=$n chains: $lib" >> $OUT/nolog.log
    timeout 200 python tools/step_timer.py --lib _ab/lib$lib.so --env 0 --n $n --steps 800 --reps 5 --nbuf 8 2>&1 | grep -v "amdgpu.ids\|ring at" >> $OUT/nolog.log
  done
done
cat $OUT/nolog.log
