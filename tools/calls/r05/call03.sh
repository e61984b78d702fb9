#!/bin/bash
# round 5, call 3: the sharder again (partition rule fixed), then VERDICT r4 "next" #4: the per-step-visible shape through the engine's own queue
# (GYMRS_AQL=2: HIP's header -- acquire + release -- on every packet, HIP launches' hints) against HIP launches (0) and chains (1)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_sharded_native.py -x -q 2>&1 | tail -5 | tee gpurun_out/r05/pytest_sharded.log
L=gpurun_out/r05/visible_through_queue.log
: > $L
for env in 0 1; do for lg in 20 21; do
  echo "# env $env 2^$lg lanes, 32 action buffers (bench.py's ring), 9 repetitions of 16000 steps" >> $L
  timeout 600 python tools/step_timer.py --env $env --n $((1<<lg)) --steps 16000 --reps 9 --aql 0,1,2 --nbuf 32 2>&1 | grep -v "^ring" >> $L
done; done
echo "# env 2 2^22 lanes" >> $L
timeout 600 python tools/step_timer.py --env 2 --n $((1<<22)) --steps 4000 --reps 7 --aql 0,1,2 --nbuf 8 2>&1 | grep -v "^ring" >> $L
cat $L
# the same tests with every step_many going through the queue with a release per launch: bit-exact like the other two submissions?
GYMRS_AQL=2 timeout 1500 python -m pytest tests/test_gpu_aql_chain.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_time_limit_elision.py tests/test_gpu_reset_log.py -x -q 2>&1 | tail -8 | tee gpurun_out/r05/pytest_aql2.log
