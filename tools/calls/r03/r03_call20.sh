#!/bin/bash
set -u
OUT=gpurun_out/r03_c20
mkdir -p $OUT
L=gym-rs_amd/libgymrs_amd.so
echo "== one engine" >> $OUT/multi.log
timeout 100 python tools/step_timer.py --lib $L --env 0 --n 1048576 --steps 1000 --reps 5 --all 1 2>&1 | grep -v "amdgpu.ids\|ring at" >> $OUT/multi.log
echo "== two engines of the same library instance, alternating chains" >> $OUT/multi.log
cp $L /tmp/libcopy.so
timeout 100 python tools/step_timer.py --lib $L --lib $L --env 0 --n 1048576 --steps 1000 --reps 5 --all 1 2>&1 | grep -v "amdgpu.ids\|ring at" >> $OUT/multi.log
echo "== two engines, GYMRS_AQL=0" >> $OUT/multi.log
GYMRS_AQL=0 timeout 100 python tools/step_timer.py --lib $L --lib $L --env 0 --n 1048576 --steps 1000 --reps 5 --all 1 2>&1 | grep -v "amdgpu.ids\|ring at" >> $OUT/multi.log
echo "== two library instances (a copy of the .so), alternating chains" >> $OUT/multi.log
timeout 100 python tools/step_timer.py --lib $L --lib /tmp/libcopy.so --env 0 --n 1048576 --steps 1000 --reps 5 --all 1 2>&1 | grep -v "amdgpu.ids\|ring at" >> $OUT/multi.log
cat $OUT/multi.log
