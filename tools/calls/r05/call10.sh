#!/bin/bash
# round 5, call 10: one more difference between the HIP runtime's packets and the dispatcher's: the dispatch is 3-dimensional there (setup = 3, y = z = 1)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
L=gpurun_out/r05/visible_through_queue_dims.log
: > $L
for e in 0 16; do
  echo "# GYMRS_AQL_EXP=$e, CartPole 2^20" >> $L
  GYMRS_AQL_EXP=$e timeout 600 python tools/step_timer.py --env 0 --n $((1<<20)) --steps 16000 --reps 5 --aql 0,2,1 --nbuf 32 2>&1 | grep "us median" >> $L
done
cat $L
