#!/bin/bash
# round 4, call 19: chain objects are parked and reused now (no HSA queue is created or destroyed per engine after the first eight): the suite three
# times, then two fuzz seeds and the big-lane cases, all with the native-backtrace handler
set -u
OUT=gpurun_out/r04_c19; mkdir -p $OUT
export TMPDIR=/tmp GYMRS_TEST_SEGV_TRACE=1
python -c "import bench; print('kernel_source_sha16', bench.kernel_source_sha16())" > $OUT/status.log
for i in 1 2 3; do
  timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/suite_$i.log 2>&1; echo "suite $i rc $? $(grep -E 'passed|failed' $OUT/suite_$i.log | tail -1)" >> $OUT/status.log
done
for seed in 51 52; do
  timeout 900 python tools/fuzz_engine_vs_twin.py --cases 220 --seed $seed > $OUT/fuzz_seed$seed.log 2>&1; echo "fuzz seed $seed rc $? $(tail -1 $OUT/fuzz_seed$seed.log)" >> $OUT/status.log
done
timeout 900 python tools/fuzz_engine_vs_twin.py --cases 6 --seed 53 --min-lanes 1400000 --max-lanes 5000000 --ops 10 > $OUT/fuzz_big.log 2>&1; echo "fuzz big rc $? $(tail -1 $OUT/fuzz_big.log)" >> $OUT/status.log
GYMRS_AQL=1 timeout 300 python tools/step_timer.py --lib _ab/libr03.so --lib gym-rs_amd/libgymrs_amd.so --steps 5000 --reps 7 > $OUT/ab_r03_vs_now_aql1.log 2>&1
echo done >> $OUT/status.log
