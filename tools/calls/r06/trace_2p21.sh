#!/bin/bash
# round 6 (VERDICT r5 "next" #5a): per-wave phase stamps and the chip-wide timeline of the CartPole step at 2^21 lanes next to 2^20 (tools/trace on a
# -DGYMRS_TRACE_TIMES build), HIP launches and chains; and the step against the in-place copy of its footprint at those sizes on this box
mkdir -p gpurun_out
for lg in 20 21; do for aql in 0 1; do
  GYMRS_AQL=$aql timeout 120 tools/trace 3 13 $lg 600 > gpurun_out/r06_trace_2p${lg}_aql$aql.log 2>&1; echo "trace 2^$lg aql=$aql rc=$?"
done; done
for aql in 0 1; do GYMRS_AQL=$aql timeout 600 python tools/size_sweep.py --env 0 --sizes 20,21,22 > gpurun_out/r06_size_sweep_2p21_aql$aql.log 2>&1; done
tail -5 gpurun_out/r06_size_sweep_2p21_aql0.log
