#!/bin/bash
# round 4, call 2: the GPU suite on the new build (XCD assert in every chain launch, dispatcher set up at engine creation), then A/B runs
set -u
OUT=gpurun_out/r04_c2; mkdir -p $OUT; REPO=$(pwd)
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/status.log
# 2. what the XCD assert costs: round 3's library against this one, alternately, chains and HIP launches
for aql in 1 0; do
  GYMRS_AQL=$aql timeout 300 python tools/step_timer.py --lib _ab/libr03.so --lib gym-rs_amd/libgymrs_amd.so --steps 5000 --reps 7 --all 1 > $OUT/ab_r03_vs_now_aql$aql.log 2>&1
done
# 3. memory hints per class of access over sizes (GYMRS_EXP_HINTS: 1 state loads, 2 state stores, 4 action loads, 8 outputs)
for lg in 21 22 23 24 25; do
  steps=$(( 6000 >> (lg - 20) ))
  for aql in 1 0; do
    GYMRS_AQL=$aql timeout 600 python tools/step_timer.py --lib _ab/libh0.so --lib _ab/libh2.so --lib _ab/libh8.so --lib _ab/libh10.so --lib _ab/libh11.so --lib _ab/libh14.so --lib _ab/libh15.so \
      --n $((1 << lg)) --steps $steps --reps 5 --nbuf 8 > $OUT/hints_2p${lg}_aql$aql.log 2>&1
  done
done
# 4. written-through (sc1) stores under HIP launches at 2^20 lanes: state stores (2), outputs (8), both (10)
GYMRS_AQL=0 timeout 300 python tools/step_timer.py --lib gym-rs_amd/libgymrs_amd.so --lib _ab/libsc2.so --lib _ab/libsc8.so --lib _ab/libsc10.so --steps 5000 --reps 7 > $OUT/sc1_hip_2p20.log 2>&1
GYMRS_AQL=1 timeout 300 python tools/step_timer.py --lib gym-rs_amd/libgymrs_amd.so --lib _ab/libsc2.so --lib _ab/libsc8.so --lib _ab/libsc10.so --steps 5000 --reps 7 > $OUT/sc1_chain_2p20.log 2>&1
# 5. the step's nine streams at other sizes
for lg in 22 23 25; do timeout 300 tools/hbm_probe $lg 2>&1 | sed -n '/^# (b)/,$p' > $OUT/hbm_probe_2p$lg.log; done
# 6. per-wave phase trace of the chain's kernel (and of the HIP-launched one)
for aql in 1 0; do GYMRS_AQL=$aql timeout 120 tools/trace 3 13 > $OUT/wave_phase_trace_aql$aql.log 2>&1; done
echo done >> $OUT/status.log
