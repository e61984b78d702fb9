#!/bin/bash
set -u
OUT=gpurun_out/r03_c7
mkdir -p $OUT
cd tools/aql
timeout 120 ./hip_ref 1048576 4000 > ../../$OUT/hip_ref.log 2>&1; cat ../../$OUT/hip_ref.log
for v in NONE AQL_PREFILLED_KERNARG AQL_QUEUE_MULTI AQL_ZERO_HINTS AQL_PROFILING; do
  echo "== variant $v" >> ../../$OUT/aql_variants.log
  env $v=1 timeout 300 ./aql_probe aql_kernels.hsaco 1048576 3000 2>&1 | grep -E "stepish alu=100, barrier|empty, barrier bit, no" >> ../../$OUT/aql_variants.log
done
cat ../../$OUT/aql_variants.log
