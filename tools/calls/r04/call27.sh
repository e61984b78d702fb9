#!/bin/bash
# round 4, call 27: do the device-wide fabric counters see lines of zeros? (tools/devcount/zero_lines.py, one counter per process)
set -u
OUT=gpurun_out/r04_c27; mkdir -p $OUT
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  ROCP_TOOL_LIBRARIES=$PWD/tools/devcount/libgymrs_devcount.so timeout 600 python tools/devcount/zero_lines.py --counters $c > $OUT/zero_lines_$c.json 2> $OUT/zero_lines_$c.err; echo "$c rc $?" >> $OUT/status.log
done
echo done >> $OUT/status.log
