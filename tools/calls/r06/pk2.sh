#!/bin/bash
# round 6, GPU call 8: tools/pkbench with more instruction forms (v_cndmask_b32 read 23 cycles in call 7: which form, which operand?), then the cold-start amplifier on
# the FINAL tree (baseline statistics, read-through read-out): fresh processes, the bench's schedule once each, chains opted in by the tool
mkdir -p gpurun_out
timeout 300 tools/pkbench > gpurun_out/r06_pk2.log 2>&1
cat gpurun_out/r06_pk2.log
OUT=gpurun_out/r06_amp5.log
: > $OUT
echo "# coldstart_amp --seconds ${1:-900}" >> $OUT
timeout 1500 python tools/coldstart_amp.py --seconds ${1:-900} 2>&1 | grep -v "resource_tracker\|warnings.warn" >> $OUT
echo "# rc=${PIPESTATUS[0]}" >> $OUT
tail -c 3000 $OUT
