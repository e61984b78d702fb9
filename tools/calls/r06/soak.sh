#!/bin/bash
# round 6: the GPU suite N times on one box, as the driver runs it (-x), plus smoke() -- the final tree must be green on a fresh box, every time
mkdir -p gpurun_out
export TMPDIR=/tmp
N=${1:-4}
OUT=gpurun_out/r06_soak_${2:-a}.log
: > $OUT
for i in $(seq 1 $N); do
    timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r06_soak_run.log 2>&1
    echo "run $i rc=$? $(grep -E 'passed|failed' gpurun_out/r06_soak_run.log | tail -1)" >> $OUT
    grep -E "^FAILED|^ERROR" gpurun_out/r06_soak_run.log >> $OUT
    if grep -qE "failed|error" gpurun_out/r06_soak_run.log; then cp gpurun_out/r06_soak_run.log gpurun_out/r06_soak_${2:-a}_failed_run$i.log; fi
done
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 >> $OUT
cat $OUT
