#!/bin/bash
# round 4, call 5: the XCD check's cost after moving its table fetch behind the state loads; several tiles per workgroup at the mid sizes
set -u
OUT=gpurun_out/r04_c5; mkdir -p $OUT; REPO=$(pwd)
export TMPDIR=/tmp
GYMRS_AQL=1 timeout 300 python tools/step_timer.py --lib _ab/libr03.so --lib gym-rs_amd/libgymrs_amd.so --steps 5000 --reps 9 --all 1 > $OUT/ab_r03_vs_now_aql1.log 2>&1
for env in 0 2; do
  for lg in 20 21 22 23 24; do
    steps=$(( 6000 >> (lg - 20) ))
    for aql in 1 0; do
      GYMRS_AQL=$aql timeout 600 python tools/step_timer.py --env $env --lib _ab/libt1.so --lib _ab/libt2.so --lib _ab/libt4.so --lib _ab/libt2w5.so --n $((1 << lg)) --steps $steps --reps 5 --nbuf 8 > $OUT/tiles_env${env}_2p${lg}_aql$aql.log 2>&1
    done
  done
done
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/status.log
echo done >> $OUT/status.log
