#!/bin/bash
# round 3, GPU call 4: which fences does HIP put on back-to-back launches; acquire-only / release-only cost; does rocprofv3 see raw AQL dispatches
set -u
OUT=gpurun_out/r03_c4
mkdir -p $OUT
export TMPDIR=/tmp
( cd tools/aql && timeout 600 ./aql_probe aql_kernels.hsaco 1048576 4000 ) > $OUT/aql_probe.log 2>&1
tail -30 $OUT/aql_probe.log
AMD_LOG_LEVEL=4 timeout 120 python tools/step_timer.py --steps 40 --reps 1 > $OUT/hip_log.txt 2>&1
grep -i "dispatch header\|Dispatch Header\|barrier=" $OUT/hip_log.txt | sort | uniq -c | sort -rn | head -20 > $OUT/hip_headers.txt
cat $OUT/hip_headers.txt
grep -c . $OUT/hip_log.txt
grep -i "header" $OUT/hip_log.txt | head -5
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_aql -o aql -- $GRAFT_REPO_ROOT/tools/aql/aql_probe $GRAFT_REPO_ROOT/tools/aql/aql_kernels.hsaco 1048576 500 > $GRAFT_REPO_ROOT/$OUT/rocprof_aql.log 2>&1
cd $GRAFT_REPO_ROOT; ls -R $OUT/prof_aql | head; find $OUT/prof_aql -name "*stats*" | head -3 | xargs -r head -20
find $OUT/prof_aql -name "*.db" -size +8M -delete
echo "call4 done"
