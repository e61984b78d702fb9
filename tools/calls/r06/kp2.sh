#!/bin/bash
# round 6, GPU call 12: is it the ORDER (loads ahead of the argument fetch) that costs the multi-generation sizes 1-3 %, or the code the compiler makes of the new structure?
# base = the tree before; kp = loads first; kpB = the new structure with the argument fetch (waited for) ahead of the loads
mkdir -p gpurun_out
OUT=gpurun_out/r06_kp2.log
: > $OUT
L="--lib _ab/libbase.so --lib _ab/libkp.so --lib _ab/libkpB.so"
run() { echo "# env $1 2^$2 aql=$4" >> $OUT; GYMRS_AQL=$4 timeout 900 python tools/step_timer.py $L --env $1 --n $((1<<$2)) --steps $3 --reps 5 --nbuf 8 2>&1 | grep -v "amdgpu.ids\|^ring" >> $OUT; }
run 0 20 16000 0
run 0 21 6000 0
run 0 22 4000 0
run 0 23 2000 0
run 0 24 1000 0
run 1 20 16000 0
run 1 21 6000 0
run 1 22 4000 0
run 1 24 1000 0
run 2 20 16000 0
run 2 22 4000 0
run 0 20 16000 0
run 0 22 4000 1
cat $OUT
