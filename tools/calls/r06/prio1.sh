#!/bin/bash
# round 6, GPU call 10: DIFFERENT issue priorities (s_setprio) for the wavefronts that share a SIMD -- the lever against the phase-locked VALU-bound waves of
# profiles/r06_wave_phase_trace_2p21.log that no earlier round tried.  Developer variants (-DGYMRS_EXP_SETPRIO=m) against the same tree without, alternately in one process.
mkdir -p gpurun_out
OUT=gpurun_out/r06_prio1.log
: > $OUT
L=""
for v in base prio1 prio2 prio4 prio5 prio6 prio11 prio12; do L="$L --lib _ab/lib$v.so"; done
for rep in 1 2; do
echo "# CartPole 2^20, HIP launches (rep $rep)" >> $OUT
GYMRS_AQL=0 timeout 900 python tools/step_timer.py $L --env 0 --n 1048576 --steps 16000 --reps 5 --nbuf 8 2>&1 | grep -v amdgpu.ids >> $OUT
done
echo "# CartPole 2^21" >> $OUT
GYMRS_AQL=0 timeout 900 python tools/step_timer.py $L --env 0 --n 2097152 --steps 6000 --reps 5 --nbuf 8 2>&1 | grep -v amdgpu.ids >> $OUT
echo "# CartPole 2^22" >> $OUT
GYMRS_AQL=0 timeout 900 python tools/step_timer.py $L --env 0 --n 4194304 --steps 3000 --reps 3 --nbuf 8 2>&1 | grep -v amdgpu.ids >> $OUT
echo "# CartPole 2^18" >> $OUT
GYMRS_AQL=0 timeout 900 python tools/step_timer.py $L --env 0 --n 262144 --steps 16000 --reps 3 --nbuf 8 2>&1 | grep -v amdgpu.ids >> $OUT
cat $OUT
