#!/bin/bash
set -u
OUT=gpurun_out/r03_c8
mkdir -p $OUT
cd tools/aql
for v in "X=1" "AQL_BATCH=64" "AQL_DEPTH=4" "AQL_DEPTH=2"; do
  echo "== probe $v" >> ../../$OUT/probe.log
  env $v timeout 300 ./aql_probe aql_kernels.hsaco 1048576 3000 2>&1 | grep -E "stepish alu=100, barrier, (no|agent f|agent ACQ)" >> ../../$OUT/probe.log
done
echo "== probe, kernel built with kernarg preload" >> ../../$OUT/probe.log
timeout 300 ./aql_probe aql_kernels_preload.hsaco 1048576 3000 2>&1 | grep -E "stepish alu=100, barrier, (no|agent f|agent ACQ)" >> ../../$OUT/probe.log
cat ../../$OUT/probe.log
cd ../..
for v in "X=1" "GYMRS_AQL_FLUSH=1" "GYMRS_AQL_DEPTH=4" "GYMRS_AQL_DEPTH=2" "GYMRS_AQL_DEPTH=16" "GYMRS_AQL_FENCES=11" "GYMRS_AQL_FENCES=11 GYMRS_AQL_DEPTH=2"; do
  echo "== engine $v" >> $OUT/engine.log
  env $v timeout 300 python tools/step_timer.py --env 0 --n 1048576 --steps 2000 --reps 7 2>&1 | grep -v "amdgpu.ids\|ring at" >> $OUT/engine.log
done
cat $OUT/engine.log
