#!/bin/bash
# round 4, call 10: the XCD check with its table load hidden behind an uncounted asm statement, one wave per workgroup (a WRONG state in the first chain of the suite: abandoned): cost per env; then the suite
set -u
OUT=gpurun_out/r04_c10; mkdir -p $OUT
export TMPDIR=/tmp
for env in 1 0 2; do
  GYMRS_AQL=1 timeout 600 python tools/step_timer.py --env $env --lib _ab/libr03.so --lib _ab/libx0.so --lib gym-rs_amd/libgymrs_amd.so --steps 5000 --reps 9 > $OUT/ab_env${env}_chain.log 2>&1
  GYMRS_AQL=0 timeout 600 python tools/step_timer.py --env $env --lib _ab/libr03.so --lib _ab/libx0.so --lib gym-rs_amd/libgymrs_amd.so --steps 5000 --reps 9 > $OUT/ab_env${env}_hip.log 2>&1
done
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/status.log
echo done >> $OUT/status.log
