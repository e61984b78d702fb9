#!/bin/bash
# round 3, GPU call 5: the engine's own AQL dispatcher: parity tests, then timing against HIP launches
set -u
OUT=gpurun_out/r03_c5
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_aql_chain.py -x -q -s 2>&1 | tail -25 > $OUT/pytest_aql.log
cat $OUT/pytest_aql.log
for env in 0 1 2; do
  for n in 1048576 4194304; do
    for a in 1 0; do
      echo "== env $env n $n GYMRS_AQL=$a" >> $OUT/timing.log
      GYMRS_AQL=$a timeout 300 python tools/step_timer.py --env $env --n $n --steps 2000 --reps 7 2>&1 | grep -v "amdgpu.ids\|ring at" >> $OUT/timing.log
    done
  done
done
cat $OUT/timing.log
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $OUT/pytest_all.log
cat $OUT/pytest_all.log
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 1 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err
echo "call5 done"
