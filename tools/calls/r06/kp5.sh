#!/bin/bash
# round 6, GPU call 19: Pendulum's step kernel with its loads ahead of the argument fetch too (`kpD` = the tree to ship) against the round-5 kernels (`base`), then ONE
# tools/refresh_evidence.sh r06 pass from this tree
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/r06_kp5.log
: > $OUT
L="--lib _ab/libbase.so --lib _ab/libkpD.so"
run() { echo "# env $1 2^$2 nbuf $4" >> $OUT; GYMRS_AQL=0 timeout 900 python tools/step_timer.py $L --env $1 --n $((1<<$2)) --steps $3 --reps 5 --nbuf $4 2>&1 | grep -v "amdgpu.ids\|^ring" >> $OUT; }
run 2 22 3000 32
run 2 22 3000 8
run 2 20 12000 8
run 0 20 16000 8
run 1 20 16000 8
run 1 22 4000 8
cat $OUT
bash tools/refresh_evidence.sh r06 > gpurun_out/r06_refresh.log 2>&1
tail -2 gpurun_out/r06_refresh.log; cat gpurun_out/r06_pytest_gpu.log gpurun_out/r06_sha.txt; cat gpurun_out/r06_bench_driver_form.json | cut -c1-700
