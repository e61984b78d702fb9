#!/bin/bash
# round 5, call 9: is it the sleeping waiter kernel on the stream's queue?  GYMRS_AQL=2 with the SYNCHRONOUS hand-over (the host waits: nothing is resident on the
# HIP stream's queue while the queue's launches run) against the asynchronous one, and --wall (host clock around call + sync) since events bracket differently
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
L=gpurun_out/r05/visible_through_queue_handover.log
: > $L
for env in 0 1; do
for ho in kernel sync; do
  echo "# env $env 2^20 lanes, GYMRS_AQL_HANDOVER=$ho, stream events" >> $L
  GYMRS_AQL_HANDOVER=$ho timeout 600 python tools/step_timer.py --env $env --n $((1<<20)) --steps 16000 --reps 7 --aql 0,2,1 --nbuf 32 2>&1 | grep "us median" >> $L
  echo "# env $env 2^20 lanes, GYMRS_AQL_HANDOVER=$ho, host wall clock around call + sync" >> $L
  GYMRS_AQL_HANDOVER=$ho timeout 600 python tools/step_timer.py --env $env --n $((1<<20)) --steps 16000 --reps 7 --aql 0,2,1 --nbuf 32 --wall 1 2>&1 | grep "us median" >> $L
done; done
cat $L
