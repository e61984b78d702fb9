#!/bin/bash
# round 5: soak of the GPU suite (VERDICT r4 "next" #7: find the 1-in-50 segmentation fault of round 4 or bound it).  usage: soak.sh TAG RUNS [ENV=VAL ...] -- pytest-args
# One line per run (rc, passed/failed counts, seconds); the full output of a run is kept only when it failed.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
TAG=$1; RUNS=$2; shift 2
ENVS=()
while [ "$1" != "--" ]; do ENVS+=("$1"); shift; done; shift
L=gpurun_out/r05/soak_$TAG.log
echo "# soak $TAG: $RUNS runs of: env ${ENVS[*]} python -m pytest $*" > $L
fails=0
for i in $(seq 1 $RUNS); do
  t0=$(date +%s)
  env "${ENVS[@]}" timeout 900 python -m pytest "$@" -q -p no:cacheprovider > /tmp/soak_run.out 2>&1
  rc=$?
  t1=$(date +%s)
  echo "run $i rc=$rc $((t1-t0))s $(tail -1 /tmp/soak_run.out)" >> $L
  if [ $rc -ne 0 ]; then fails=$((fails+1)); cp /tmp/soak_run.out gpurun_out/r05/soak_${TAG}_fail_$i.out; dmesg 2>/dev/null | tail -5 >> $L; fi
done
echo "# $TAG: $fails of $RUNS runs failed" >> $L
tail -3 $L
