#!/bin/bash
# round 6, GPU call 5: VERDICT r5 "next" #2 at scale -- every stream <-> chain hand-over class per iteration (handover_amp.py --audit), 8 processes on the one GPU,
# HIP launches (the product's default) and opt-in chains, flag sets 3 and 7; then the GPU suite, smoke() and the driver's bench command on this tree.
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/r06_audit1.log
: > $OUT
run() { echo "# $*" >> $OUT; timeout 900 python tools/handover_amp.py "$@" 2>&1 | grep -v "^/usr/lib/python3.10/multiprocessing\|^  warnings.warn" >> $OUT; echo "# rc=${PIPESTATUS[0]}" >> $OUT; }
run --procs 8 --seconds 150 --mode hip   --audit --flags 7
run --procs 8 --seconds 150 --mode both  --audit --flags 7 --handover kernel
run --procs 8 --seconds 120 --mode chain --audit --flags 7 --handover auto
run --procs 8 --seconds 90  --mode both  --audit --flags 3 --handover kernel --lockstep
run --procs 8 --seconds 90  --mode hip   --audit --flags 3 --check-clear
tail -c 6000 $OUT
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r06_pytest_gpu_call5.log 2>&1
echo "rc=$?" >> gpurun_out/r06_pytest_gpu_call5.log
grep -n "^FAILED\|passed\|failed" gpurun_out/r06_pytest_gpu_call5.log | tail -20
timeout 300 python __graft_entry__.py smoke > gpurun_out/r06_smoke_call5.log 2>&1; tail -3 gpurun_out/r06_smoke_call5.log
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --full-out gpurun_out/r06_bench_driver_form_call5_full.json > gpurun_out/r06_bench_driver_form_call5.json 2> gpurun_out/r06_bench_driver_form_call5.err
cat gpurun_out/r06_bench_driver_form_call5.json
