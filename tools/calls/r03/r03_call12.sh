#!/bin/bash
set -u
OUT=gpurun_out/r03_c12
mkdir -p $OUT
for lib in vbase vplain vnolog; do
  for v in "GYMRS_AQL=0" "GYMRS_AQL_LAZY_WAIT=1" "GYMRS_AQL_LAZY_WAIT=1 GYMRS_AQL_FENCES=11"; do
    echo "== $lib $v" >> $OUT/ablate.log
    env $v timeout 100 python tools/step_timer.py --lib _ab/lib$lib.so --env 0 --n 1048576 --steps 2000 --reps 8 --all 1 --wall 1 2>&1 | grep -v "amdgpu.ids\|ring at" >> $OUT/ablate.log
  done
done
cat $OUT/ablate.log
