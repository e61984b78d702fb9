#!/bin/bash
# round 6, GPU call 7: what does a packed f32 VALU instruction cost on this part with the SIMD as full as the step kernel keeps it (tools/pkbench), and the
# step kernel built without the SLP vectoriser (no v_pk_*, no packing moves) against the product's flags, timed alternately in one process (tools/step_timer.py)
mkdir -p gpurun_out
OUT=gpurun_out/r06_pk1.log
: > $OUT
echo "# tools/pkbench" >> $OUT
timeout 120 tools/pkbench >> $OUT 2>&1
for rep in 1 2; do
echo "# step_timer CartPole 2^20, base vs noslp, HIP launches (rep $rep)" >> $OUT
GYMRS_AQL=0 timeout 600 python tools/step_timer.py --lib _ab/libbase.so --lib _ab/libnoslp.so --env 0 --n 1048576 --steps 16000 --reps 7 --nbuf 8 2>&1 | grep -v amdgpu.ids >> $OUT
done
echo "# step_timer CartPole 2^21 / 2^22" >> $OUT
GYMRS_AQL=0 timeout 600 python tools/step_timer.py --lib _ab/libbase.so --lib _ab/libnoslp.so --env 0 --n 2097152 --steps 6000 --reps 5 --nbuf 8 2>&1 | grep -v amdgpu.ids >> $OUT
GYMRS_AQL=0 timeout 600 python tools/step_timer.py --lib _ab/libbase.so --lib _ab/libnoslp.so --env 0 --n 4194304 --steps 3000 --reps 5 --nbuf 8 2>&1 | grep -v amdgpu.ids >> $OUT
cat $OUT
