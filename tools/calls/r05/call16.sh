#!/bin/bash
# call 16: soak part 3 saw ONE wrong answer in 22 full-suite runs: test_eight_ranks_share_the_gpu[32768] -- 8 processes sharing the GPU (16 queues) printed
# n_episodes 11642124 where one engine of the same lanes, and the CPU twin, give 11479920.  Which submission?  The same command with ONE call shape, N times each,
# the line's statistics against the twin's for that schedule.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
L=gpurun_out/r05/eight_ranks_repro.log
for round in 1 2 3; do for shape in both; do
  bad=0; n=0
  for i in $(seq 1 30); do
    out=$(timeout 300 python tools/exp_eight_ranks.py $shape 2>&1 | tail -1)
    n=$((n+1))
    case "$out" in OK) ;; *) bad=$((bad+1)); echo "$shape run $i: $out" >> $L;; esac
  done
  echo "# $shape (round $round): $bad of $n runs printed statistics that differ from the twin's" | tee -a $L
done; done
