#!/bin/bash
# round 5, call 4: why does the queue lose to HIP launches at 2^20 CartPole lanes with the same header, kernel and hints (6.72 vs 6.36 us)?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
L=gpurun_out/r05/visible_through_queue_why.log
: > $L
run() { echo "# $1" >> $L; shift; env "$@" timeout 600 python tools/step_timer.py --env 0 --n $((1<<20)) --steps 16000 --reps 7 --aql 0,2 --nbuf 32 2>&1 | grep -v "^ring\|amdgpu.ids" >> $L; }
run "as built" X=1
run "a completion signal on every releasing packet (GYMRS_AQL_EXP=1)" GYMRS_AQL_EXP=1
run "doorbell after every packet (GYMRS_AQL_FLUSH=1)" GYMRS_AQL_FLUSH=1
run "doorbell every 8 packets" GYMRS_AQL_FLUSH=8
run "256 work-items per workgroup for the queue's launches (GYMRS_DEV_THREADS=256)" GYMRS_DEV_THREADS=256
run "release at SYSTEM scope (GYMRS_AQL_FENCES=12: ignored by the release flag, acquire agent)" GYMRS_AQL_FENCES=12
run "a short queue: 64 packets (GYMRS_AQL_QUEUE=64)" GYMRS_AQL_QUEUE=64 GYMRS_AQL_FLUSH=1
echo "# nt hint variants through the queue: --nts 0,1,2,3 (0 auto = 1 at this size)" >> $L
timeout 600 python tools/step_timer.py --env 0 --n $((1<<20)) --steps 16000 --reps 5 --aql 2 --nts 1,2,3 --nbuf 32 2>&1 | grep -v "^ring\|amdgpu.ids" >> $L
cat $L
