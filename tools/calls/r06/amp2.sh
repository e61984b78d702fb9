#!/bin/bash
# round 6, GPU call 2: the cold-start amplifier (tools/coldstart_amp.py): fresh processes, the bench's schedule once each, post-mortem on a wrong count
mkdir -p gpurun_out
OUT=gpurun_out/r06_amp2.log
: > $OUT
echo "# coldstart_amp --seconds ${1:-480}" >> $OUT
timeout 1200 python tools/coldstart_amp.py --seconds ${1:-480} >> $OUT 2>&1
echo "# rc=$?" >> $OUT
tail -c 8000 $OUT
