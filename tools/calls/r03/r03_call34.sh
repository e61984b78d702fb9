#!/bin/bash
set -u
OUT=gpurun_out/r03_c34
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_time_limit_elision.py tests/test_gpu_aql_chain.py tests/test_gpu_reset_log.py -x -q 2>&1 | tail -6
timeout 300 python tests/soak_per_step.py 2>&1 | grep -v amdgpu.ids | tail -3
for a in 1 0; do
  echo "== cartpole 2^20 flags 7 (AUTO_RESET|TRACK_STATS|TIME_LIMIT: limit elision) GYMRS_AQL=$a" >> $OUT/flags7.log
  GYMRS_AQL=$a timeout 100 python tools/step_timer.py --env 0 --n 1048576 --flags 7 --steps 2000 --reps 7 --all 1 2>&1 | grep -v "amdgpu.ids\|ring at" >> $OUT/flags7.log
done
cat $OUT/flags7.log
timeout 100 python tools/exp_limit_elision.py 2>&1 | grep -v amdgpu.ids | tail -12
