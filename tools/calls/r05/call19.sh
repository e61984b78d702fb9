#!/bin/bash
# call 19: does the hand-over decide the 8-processes-on-one-GPU wrong count?  the test's command with the hand-over forced, 50 runs each
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
L=gpurun_out/r05/eight_ranks_handover.log
for ho in kernel; do
  bad=0; n=0
  for i in $(seq 1 110); do
    out=$(GYMRS_AQL_HANDOVER=$ho timeout 300 python tools/exp_eight_ranks.py both 2>&1 | tail -1)
    n=$((n+1))
    case "$out" in OK) ;; *) bad=$((bad+1)); echo "handover $ho run $i: $out" >> $L;; esac
  done
  echo "# GYMRS_AQL_HANDOVER=$ho: $bad of $n runs printed statistics that differ from the twin's" | tee -a $L
done
