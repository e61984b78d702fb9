#!/bin/bash
# round 3, GPU call 1: the GPU suite on the round's first changes + the slow-mode experiment (tools/exp_slow_mode.py)
set -u
OUT=gpurun_out/r03_c1
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest.log 2>&1
echo "pytest done: $(tail -1 $OUT/pytest.log)"
for i in 1 2 3 4 5 6 7 8; do
  SLOW_MODE_DETAIL=$OUT/slow_smi_$i.detail.json timeout 120 python tools/exp_slow_mode.py --seconds 6 --smi-ms 5 --tag smi_$i >> $OUT/slow_smi.jsonl 2>> $OUT/slow.err
done
for i in 1 2 3 4 5 6; do
  timeout 120 python tools/exp_slow_mode.py --seconds 6 --smi-ms 0 --tag nosmi_$i >> $OUT/slow_nosmi.jsonl 2>> $OUT/slow.err
done
for i in 1 2 3 4; do
  timeout 120 python tools/exp_slow_mode.py --seconds 6 --smi-ms 0 --probe-every 0 --tag plain_$i >> $OUT/slow_plain.jsonl 2>> $OUT/slow.err
done
for i in 1 2 3 4 5 6; do
  timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-probe >> $OUT/bench_driver_form.jsonl 2>> $OUT/bench.err
done
amd-smi static --asic --limit 2>&1 | head -60 > $OUT/amdsmi_static.txt
amd-smi metric --clock --power 2>&1 | head -80 > $OUT/amdsmi_metric.txt
echo "call1 done"
