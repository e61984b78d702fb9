#!/bin/bash
# round 3, GPU call 2: raw-AQL launch-boundary probe; per-array memory hints A/B over sizes; a slow-mode sample of this box
set -u
OUT=gpurun_out/r03_c2
mkdir -p $OUT
export TMPDIR=/tmp
( cd tools/aql && timeout 600 ./aql_probe aql_kernels.hsaco 1048576 4000 ) > $OUT/aql_probe.log 2>&1
echo "aql rc=$?"; tail -25 $OUT/aql_probe.log
( cd tools/aql && timeout 300 ./aql_probe aql_kernels.hsaco 4194304 1500 ) > $OUT/aql_probe_2p22.log 2>&1
LIBS="--lib _ab/libbase.so --lib _ab/libh12.so --lib _ab/libh14.so --lib _ab/libh13.so --lib _ab/libh4.so --lib _ab/libh8.so"
for n in 1048576 2097152 4194304 8388608; do
  echo "== cartpole n=$n auto hint (base) vs GYMRS_EXP_HINTS variants" >> $OUT/hints.log
  timeout 600 python tools/step_timer.py $LIBS --env 0 --n $n --steps 1500 --reps 5 >> $OUT/hints.log 2>&1
  for nt in 1 2; do
    echo "== cartpole n=$n base lib, --nt $nt (1 = all hinted, 2 = all plain)" >> $OUT/hints.log
    timeout 300 python tools/step_timer.py --lib _ab/libbase.so --env 0 --n $n --steps 1500 --reps 5 --nt $nt >> $OUT/hints.log 2>&1
  done
done
echo "== pendulum n=4194304" >> $OUT/hints.log
timeout 600 python tools/step_timer.py $LIBS --env 2 --n 4194304 --steps 1000 --reps 5 >> $OUT/hints.log 2>&1
for nt in 1 2; do
  echo "== pendulum n=4194304 base lib --nt $nt" >> $OUT/hints.log
  timeout 300 python tools/step_timer.py --lib _ab/libbase.so --env 2 --n 4194304 --steps 1000 --reps 5 --nt $nt >> $OUT/hints.log 2>&1
done
for n in 1048576 4194304; do
  echo "== mountain_car n=$n" >> $OUT/hints.log
  timeout 600 python tools/step_timer.py $LIBS --env 1 --n $n --steps 1500 --reps 5 >> $OUT/hints.log 2>&1
done
SLOW_MODE_DETAIL=$OUT/slow.detail.json timeout 120 python tools/exp_slow_mode.py --seconds 4 --smi-ms 5 --tag c2 > $OUT/slow.jsonl 2>> $OUT/slow.err
echo "call2 done"
