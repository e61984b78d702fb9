#!/bin/bash
# round 6, GPU call 14: CartPole's step kernel with its loads ahead of the argument fetch (`kpC`; MountainCar / Pendulum kernels byte-identical to before) against the
# tree before (`base`), then ONE tools/refresh_evidence.sh r06 pass from this tree (suite, smoke, traffic, bench lines, kernel trace, PMC, sweeps)
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/r06_kp4.log
: > $OUT
L="--lib _ab/libbase.so --lib _ab/libkpC.so"
run() { echo "# env $1 2^$2 aql=$4" >> $OUT; GYMRS_AQL=$4 timeout 900 python tools/step_timer.py $L --env $1 --n $((1<<$2)) --steps $3 --reps 5 --nbuf 8 2>&1 | grep -v "amdgpu.ids\|^ring" >> $OUT; }
run 0 20 16000 0
run 0 16 16000 0
run 0 18 16000 0
run 0 21 6000 0
run 0 22 4000 0
run 0 23 2000 0
run 0 24 1000 0
run 0 25 500 0
run 0 20 16000 1
run 0 20 16000 0
run 1 20 16000 0
cat $OUT
bash tools/refresh_evidence.sh r06 > gpurun_out/r06_refresh.log 2>&1
tail -2 gpurun_out/r06_refresh.log; cat gpurun_out/r06_pytest_gpu.log gpurun_out/r06_sha.txt; cat gpurun_out/r06_bench_driver_form.json | cut -c1-900
