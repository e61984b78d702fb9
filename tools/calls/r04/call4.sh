#!/bin/bash
# round 4, call 4: the GPU suite (XCD table recorded per chain), the chain's A/B against round 3, chain hints by size, traces, bench, device-wide traffic
set -u
OUT=gpurun_out/r04_c4; mkdir -p $OUT; REPO=$(pwd)
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/status.log
GYMRS_AQL=1 timeout 300 python tools/step_timer.py --lib _ab/libr03.so --lib gym-rs_amd/libgymrs_amd.so --steps 5000 --reps 9 --all 1 > $OUT/ab_r03_vs_now_aql1.log 2>&1
for env in 0 1 2; do
  for lg in 20 21 22 23 24 25; do
    [ $env = 2 ] && [ $lg = 25 ] && continue
    steps=$(( 6000 >> (lg - 20) ))
    GYMRS_AQL=1 timeout 600 python tools/step_timer.py --env $env --n $((1 << lg)) --steps $steps --reps 5 --nbuf 8 --nts 0,1,2,3 > $OUT/hints_env${env}_2p${lg}_aql1.log 2>&1
  done
done
for aql in 1 0; do GYMRS_AQL=$aql timeout 120 tools/trace 3 13 > $OUT/wave_phase_trace_aql$aql.log 2>&1; done
timeout 900 python tools/devcount/collect.py --out $OUT/devcount_traffic.json > $OUT/devcount_collect.log 2>&1; echo "devcount rc $?" >> $OUT/status.log
timeout 900 python bench.py --cpu-seconds 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?" >> $OUT/status.log
echo done >> $OUT/status.log
