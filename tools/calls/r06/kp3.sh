#!/bin/bash
# round 6, GPU call 13: the switched kernel (`kpS`: loads ahead of the argument fetch for launches of one generation, the fetch first for longer ones -- bit 63 of the
# n_fast word) against the tree before (`base`), alternately in one process, all three envs over sizes; then the FULL GPU suite and smoke() on the product build
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/r06_kp3.log
: > $OUT
L="--lib _ab/libbase.so --lib _ab/libkpS.so"
run() { echo "# env $1 2^$2 aql=$4" >> $OUT; GYMRS_AQL=$4 timeout 900 python tools/step_timer.py $L --env $1 --n $((1<<$2)) --steps $3 --reps 5 --nbuf 8 2>&1 | grep -v "amdgpu.ids\|^ring" >> $OUT; }
for env in 0 1 2; do
run $env 16 16000 0
run $env 18 16000 0
run $env 20 16000 0
run $env 21 6000 0
run $env 22 4000 0
run $env 24 1000 0
done
run 0 20 16000 1
run 1 20 16000 1
run 0 20 16000 0
cat $OUT
timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r06_kp3_pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r06_kp3_pytest.log
grep -n "^FAILED\|^ERROR\|passed\|failed\|rc=" gpurun_out/r06_kp3_pytest.log | tail
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
