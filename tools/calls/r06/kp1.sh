#!/bin/bash
# round 6, GPU call 11: the step kernel issues its state loads from the PRELOADED scalars before it fetches the rest of its arguments (`kp`) against the tree
# before (`base`), alternately in one process; then the correctness tests on the product library built from the new source
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/r06_kp1.log
: > $OUT
L="--lib _ab/libbase.so --lib _ab/libkp.so"
for rep in 1 2; do
echo "# CartPole 2^20, HIP launches (rep $rep)" >> $OUT
GYMRS_AQL=0 timeout 900 python tools/step_timer.py $L --env 0 --n 1048576 --steps 16000 --reps 7 --nbuf 8 2>&1 | grep -v "amdgpu.ids\|^ring" >> $OUT
done
for lg in 18 21 22 24; do
echo "# CartPole 2^$lg" >> $OUT
GYMRS_AQL=0 timeout 900 python tools/step_timer.py $L --env 0 --n $((1<<lg)) --steps $((lg < 21 ? 16000 : (lg < 24 ? 5000 : 1000))) --reps 5 --nbuf 8 2>&1 | grep -v "amdgpu.ids\|^ring" >> $OUT
done
echo "# CartPole 2^20, chains (GYMRS_AQL=1)" >> $OUT
GYMRS_AQL=1 timeout 900 python tools/step_timer.py $L --env 0 --n 1048576 --steps 16000 --reps 5 --nbuf 8 2>&1 | grep -v "amdgpu.ids\|^ring" >> $OUT
for env in 1 2; do for lg in 20 22; do
echo "# env $env 2^$lg" >> $OUT
GYMRS_AQL=0 timeout 900 python tools/step_timer.py $L --env $env --n $((1<<lg)) --steps $((lg < 21 ? 16000 : 5000)) --reps 5 --nbuf 8 2>&1 | grep -v "amdgpu.ids\|^ring" >> $OUT
done; done
cat $OUT
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "not oversubscribed" --deselect tests/test_gpu_handover.py > gpurun_out/r06_kp1_pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r06_kp1_pytest.log
grep -n "^FAILED\|^ERROR\|passed\|failed\|rc=" gpurun_out/r06_kp1_pytest.log | tail
