#!/bin/bash
# round 5, call 8: VERDICT r4 "next" #5 (b) -- CartPole 2^21 lanes: 256-work-item groups (the 512 window ends below 2^21), plain loads + streamed stores (hints 10),
# state loads + outputs streamed (hints 9), and their combinations, developer builds of ONE tree timed alternately in one process against the product kernel
# (r05_base = the same sources with no extra flag); HIP launches and chains; 2^20 and 2^22 beside it so that nothing else regresses.  Then the new parity test.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
L=gpurun_out/r05/cartpole_2p21.log
: > $L
LIBS="--lib _ab/libr05_base.so --lib _ab/libr05_b256.so --lib _ab/libr05_h10.so --lib _ab/libr05_b256_h10.so --lib _ab/libr05_h9.so --lib _ab/libr05_b256_h9.so"
for lg in 21 20 22; do for q in 0 1; do
  echo "# 2^$lg CartPole lanes, GYMRS_AQL=$q ($([ $q = 0 ] && echo 'HIP launches: per-step visible' || echo chains)), 8 action buffers, 7 repetitions of 6000 steps" >> $L
  GYMRS_AQL=$q timeout 900 python tools/step_timer.py $LIBS --env 0 --n $((1<<lg)) --steps 6000 --reps 7 --nbuf 8 2>&1 | grep "us median" >> $L
done; done
echo "# 2^21 lanes again with bench.py's ring of 32 action buffers" >> $L
for q in 0 1; do
  echo "# GYMRS_AQL=$q" >> $L
  GYMRS_AQL=$q timeout 900 python tools/step_timer.py $LIBS --env 0 --n $((1<<21)) --steps 6000 --reps 7 --nbuf 32 2>&1 | grep "us median" >> $L
done
cat $L
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_envs.py -x -q -k "golden or example or cpp" 2>&1 | tail -4 | tee gpurun_out/r05/pytest_call08.log
