#!/bin/bash
# round 6: Pendulum with its loads ahead of the argument fetch (`kp`, the all-env variant of call 11) against the by-value kernels (`base`), at its BASELINE size with both ring sizes
mkdir -p gpurun_out
OUT=gpurun_out/r06_pd1_${1:-a}.log
: > $OUT
L="--lib _ab/libbase.so --lib _ab/libkp.so"
run() { echo "# env 2 2^$1 nbuf $3" >> $OUT; GYMRS_AQL=0 timeout 900 python tools/step_timer.py $L --env 2 --n $((1<<$1)) --steps $2 --reps 7 --nbuf $3 2>&1 | grep -v "amdgpu.ids\|^ring" >> $OUT; }
run 22 3000 32
run 22 3000 8
run 22 3000 32
run 22 3000 8
run 20 12000 8
run 21 6000 8
run 23 1500 8
cat $OUT
