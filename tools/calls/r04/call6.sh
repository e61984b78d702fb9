#!/bin/bash
# round 4, call 6: what the XCD check costs now; suite; bench
set -u
OUT=gpurun_out/r04_c6; mkdir -p $OUT
export TMPDIR=/tmp
GYMRS_AQL=1 timeout 300 python tools/step_timer.py --lib _ab/libr03.so --lib gym-rs_amd/libgymrs_amd.so --steps 5000 --reps 9 --all 1 > $OUT/ab_r03_vs_now_aql1.log 2>&1
GYMRS_AQL=1 GYMRS_DEV_NO_XCC_CHECK=1 timeout 300 python tools/step_timer.py --lib _ab/libr03.so --lib gym-rs_amd/libgymrs_amd.so --steps 5000 --reps 9 --all 1 > $OUT/ab_r03_vs_now_nocheck_aql1.log 2>&1
for env in 1 2; do GYMRS_AQL=1 timeout 300 python tools/step_timer.py --env $env --lib _ab/libr03.so --lib gym-rs_amd/libgymrs_amd.so --steps 5000 --reps 7 > $OUT/ab_r03_vs_now_env${env}_aql1.log 2>&1; done
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/status.log
timeout 900 python bench.py --cpu-seconds 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?" >> $OUT/status.log
echo done >> $OUT/status.log
