#!/bin/bash
set -u
OUT=gpurun_out/r03_c9
mkdir -p $OUT
cd tools/aql
for name in fatargs fatcode fatboth; do
  echo "== $name: HIP" >> ../../$OUT/fat.log
  timeout 60 ./hip_ref_$name 1048576 3000 2>&1 | grep "alu=100" | tail -2 >> ../../$OUT/fat.log
  echo "== $name: raw AQL" >> ../../$OUT/fat.log
  timeout 120 ./aql_probe_$name aql_kernels_$name.hsaco 1048576 3000 2>&1 | grep -E "stepish alu=100, barrier, (no|agent fences|agent ACQ)" >> ../../$OUT/fat.log
done
cat ../../$OUT/fat.log
cd ../..
timeout 100 python tools/step_timer.py --env 0 --n 1048576 --steps 2000 --reps 9 --all 1 2>&1 | grep -v "amdgpu.ids\|ring at" > $OUT/engine_reps.log
cat $OUT/engine_reps.log
