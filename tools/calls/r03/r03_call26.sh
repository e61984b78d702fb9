#!/bin/bash
set -u
OUT=gpurun_out/r03_c26
mkdir -p $OUT
for n in 1048576 2097152 4194304 8388608 16777216; do
  for lib in wbase w7 w8; do
    echo "== cartpole n=$n chains: $lib" >> $OUT/occ.log
    timeout 200 python tools/step_timer.py --lib _ab/lib$lib.so --env 0 --n $n --steps 1000 --reps 5 2>&1 | grep -v "amdgpu.ids\|ring at" >> $OUT/occ.log
  done
done
cat $OUT/occ.log
