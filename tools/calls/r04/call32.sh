#!/bin/bash
# round 4, call 32: CartPole's reward-store elision in the product (from 128 MiB per step on): the GPU suite, big-lane fuzz cases where it is active, timings
set -u
OUT=gpurun_out/r04_c32; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $? $(grep -E 'passed|failed' $OUT/pytest_gpu.log | tail -1)" >> $OUT/status.log
timeout 900 python tools/fuzz_engine_vs_twin.py --cases 8 --seed 71 --min-lanes 3300000 --max-lanes 5000000 --ops 10 > $OUT/fuzz_big.log 2>&1; echo "fuzz big rc $? $(tail -1 $OUT/fuzz_big.log)" >> $OUT/status.log
GYMRS_DEV_ELIDE_REWARD=1 timeout 900 python tools/fuzz_engine_vs_twin.py --cases 120 --seed 72 > $OUT/fuzz_forced.log 2>&1; echo "fuzz forced-on rc $? $(tail -1 $OUT/fuzz_forced.log)" >> $OUT/status.log
for aql in 0 1; do for lg in 22 24; do
  GYMRS_AQL=$aql timeout 300 python tools/step_timer.py --n $((1 << lg)) --steps $(( 6000 >> (lg - 20) )) --reps 5 --nbuf 8 2>&1 | grep median | head -1 | sed "s/^/aql $aql 2^$lg: /" >> $OUT/status.log
done; done
echo done >> $OUT/status.log
