#!/bin/bash
# round 4, call 26: a chain whose launches each carry an agent-scope RELEASE (GYMRS_AQL_FENCES=11) against the chain as it is (release at the end only) and
# against HIP launches, by size: at 2^21 CartPole lanes the chain is SLOWER than HIP launches (12.4 vs 11.6 us) -- an L2 full of dirty lines?
set -u
OUT=gpurun_out/r04_c26; mkdir -p $OUT
export TMPDIR=/tmp
for env in 0 1 2; do
  for lg in 20 21 22 23 24; do
    steps=$(( 6000 >> (lg - 20) ))
    for rep in 1 2; do
      a=$(GYMRS_AQL=1 timeout 300 python tools/step_timer.py --env $env --n $((1 << lg)) --steps $steps --reps 5 --nbuf 8 2>&1 | grep median | head -1)
      b=$(GYMRS_AQL=1 GYMRS_AQL_FENCES=11 timeout 300 python tools/step_timer.py --env $env --n $((1 << lg)) --steps $steps --reps 5 --nbuf 8 2>&1 | grep median | head -1)
      c=$(GYMRS_AQL=0 timeout 300 python tools/step_timer.py --env $env --n $((1 << lg)) --steps $steps --reps 5 --nbuf 8 2>&1 | grep median | head -1)
      echo "env $env 2^$lg lanes | chain: $a" >> $OUT/release_per_launch.log
      echo "env $env 2^$lg lanes | chain, release on every launch: $b" >> $OUT/release_per_launch.log
      echo "env $env 2^$lg lanes | HIP launches: $c" >> $OUT/release_per_launch.log
    done
  done
done
echo done >> $OUT/status.log
