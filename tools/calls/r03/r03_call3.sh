#!/bin/bash
# round 3, GPU call 3: AQL probe with device-resident kernel arguments; action-ring depth vs the mid-size gap; the new bench.py
set -u
OUT=gpurun_out/r03_c3
mkdir -p $OUT
export TMPDIR=/tmp
( cd tools/aql && timeout 600 ./aql_probe aql_kernels.hsaco 1048576 4000 ) > $OUT/aql_probe.log 2>&1
echo "aql rc=$?"; tail -28 $OUT/aql_probe.log
for n in 2097152 4194304; do
  for nbuf in 2 4 8 16 32; do
    echo "== cartpole n=$n nbuf=$nbuf (auto hint = plain at these sizes)" >> $OUT/ring_depth.log
    timeout 300 python tools/step_timer.py --lib _ab/libbase.so --lib _ab/libh8.so --env 0 --n $n --steps 1500 --reps 5 --nbuf $nbuf 2>&1 | grep -v "amdgpu.ids\|ring at" >> $OUT/ring_depth.log
  done
done
for nbuf in 2 8 32; do
  echo "== pendulum n=4194304 nbuf=$nbuf" >> $OUT/ring_depth.log
  timeout 300 python tools/step_timer.py --lib _ab/libbase.so --lib _ab/libh13.so --env 2 --n 4194304 --steps 1000 --reps 5 --nbuf $nbuf 2>&1 | grep -v "amdgpu.ids\|ring at" >> $OUT/ring_depth.log
done
python tools/size_sweep.py > $OUT/size_sweep.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err
timeout 600 python bench.py --cpu-seconds 2 > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 900 python bench.py --gpus 8 --oversubscribe --n-envs 65536 --steps 50 --warmup 10 --cpu-seconds 1 > $OUT/bench_over8.json 2> $OUT/bench_over8.err
SLOW_MODE_DETAIL=$OUT/slow.detail.json timeout 120 python tools/exp_slow_mode.py --seconds 4 --smi-ms 5 --tag c3 > $OUT/slow.jsonl 2>> $OUT/slow.err
echo "call3 done"
