#!/bin/bash
# round 4, calls 30-32 (one box each): the driver's own command three times in a row, with the launching thread's enqueue time per launch on the line --
# if a box shows the 6.8 us mode of evidence run r04d, host_enqueue_us_per_step says whether the thread or the device sets the pace
set -u
OUT=gpurun_out/r04_c30_$1; mkdir -p $OUT
export TMPDIR=/tmp
for i in 1 2 3; do
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 1 --no-configs > $OUT/driver_form_$i.json 2> $OUT/driver_form_$i.err; echo "run $i rc $?" >> $OUT/status.log
done
lscpu | grep -E "Model name|^CPU\(s\)|NUMA node\(s\)" >> $OUT/status.log
echo done >> $OUT/status.log
