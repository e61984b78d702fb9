#!/bin/bash
# round 4, call 1: experiments only (no product change yet)
#  1. visible_probe: a release-free VISIBLE step?
#  2. hbm_probe: what a pure memory kernel gets from HBM, as a plain copy and in the step's nine streams (next #4)
#  3. device-wide counters around a free-running chain (next #1c)
#  4. per-wave phase trace of the CHAIN kernel (next #7)
#  5. the bench line of this box with the round-3 library (reference for everything that follows)
set -u
OUT=gpurun_out/r04_c1; mkdir -p $OUT; REPO=$(pwd)
export TMPDIR=/tmp
( cd tools/aql && timeout 300 ./visible_probe visible_kernels.hsaco 1048576 3000 ) > $OUT/visible_probe.log 2>&1
( cd tools/aql && VIS_ALU=0 timeout 300 ./visible_probe visible_kernels.hsaco 1048576 3000 ) > $OUT/visible_probe_alu0.log 2>&1
timeout 600 tools/hbm_probe 24 > $OUT/hbm_probe.log 2>&1
for ctr in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ,TCC_EA0_RDREQ_32B,TCC_BUBBLE TCC_EA0_WRREQ,TCC_EA0_WRREQ_64B; do
  tag=$(echo $ctr | tr ',' '+')
  ROCP_TOOL_LIBRARIES=$REPO/tools/devcount/libgymrs_devcount.so timeout 300 python tools/devcount/chain_traffic.py --counters $ctr > $OUT/devcount_$tag.json 2> $OUT/devcount_$tag.err
  echo "devcount $ctr rc $?" >> $OUT/status.log
done
# the same script without the tool: is the chain's speed disturbed by the tool's presence?
timeout 300 python tools/step_timer.py --steps 20000 --reps 5 > $OUT/step_timer_plain.log 2>&1
for aql in 1 0; do
  GYMRS_AQL=$aql timeout 120 tools/trace 3 13 > $OUT/wave_phase_trace_aql$aql.log 2>&1
done
timeout 600 python bench.py --cpu-seconds 2 > $OUT/bench.json 2> $OUT/bench.err
echo done >> $OUT/status.log
