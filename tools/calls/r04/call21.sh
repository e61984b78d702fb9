#!/bin/bash
# round 4, call 21: the convoy question (mid-size regime).  (c) of tools/hbm_probe: pure delay vs FMAs, first generation staggered; the real kernel with
# GYMRS_EXP_STAGGER (HIP launches and chains); per-wave timelines of XCD 0 at 2^22 lanes, plain and staggered
set -u
OUT=gpurun_out/r04_c21; mkdir -p $OUT
export TMPDIR=/tmp
for lg in 21 22 23; do timeout 300 tools/hbm_probe $lg phase > $OUT/phase_probe_2p$lg.log 2>&1; echo "probe 2^$lg rc $?" >> $OUT/status.log; done
for aql in 0 1; do
  for lg in 21 22 23 24; do
    steps=$(( 6000 >> (lg - 20) ))
    GYMRS_AQL=$aql timeout 600 python tools/step_timer.py --lib _ab/libbase.so --lib _ab/libstag2.so --lib _ab/libstag4.so --lib _ab/libstag8.so --n $((1 << lg)) --steps $steps --reps 7 --nbuf 8 > $OUT/stagger_2p${lg}_aql$aql.log 2>&1
    echo "stagger 2^$lg aql $aql rc $?" >> $OUT/status.log
  done
done
for lg in 22 23; do
  GYMRS_AQL=0 timeout 120 tools/trace 3 13 $lg 600 > $OUT/timeline_2p${lg}_plain.log 2>&1; echo "timeline rc $?" >> $OUT/status.log
  GYMRS_AQL=0 LD_LIBRARY_PATH=_ab/stag4 timeout 120 tools/trace 3 13 $lg 600 > $OUT/timeline_2p${lg}_stag4.log 2>&1; echo "timeline stag rc $?" >> $OUT/status.log
done
GYMRS_AQL=0 timeout 120 tools/trace 3 13 20 600 > $OUT/timeline_2p20_plain.log 2>&1
echo done >> $OUT/status.log
