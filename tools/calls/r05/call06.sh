#!/bin/bash
# round 5, call 6: the chain test again (dev hook exported), queue properties the HIP runtime's own queues have, 8 lanes per work-item at 2^20, in-process bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_aql_chain.py -x -q 2>&1 | tail -4 | tee gpurun_out/r05/pytest_call06.log
L=gpurun_out/r05/visible_through_queue_why2.log
: > $L
run() { echo "# $1" >> $L; shift; env "$@" timeout 600 python tools/step_timer.py --env 0 --n $((1<<20)) --steps 16000 --reps 5 --aql 0,2 --nbuf 32 2>&1 | grep -v "^ring\|amdgpu.ids" >> $L; }
run "as built" X=1
run "MULTI-producer queue (GYMRS_AQL_EXP=2)" GYMRS_AQL_EXP=2
run "high-priority queue (GYMRS_AQL_EXP=4)" GYMRS_AQL_EXP=4
run "profiling enabled on the queue (GYMRS_AQL_EXP=8)" GYMRS_AQL_EXP=8
run "profiling + a completion signal per packet (GYMRS_AQL_EXP=9)" GYMRS_AQL_EXP=9
run "all four (GYMRS_AQL_EXP=15)" GYMRS_AQL_EXP=15
echo "# 8 lanes per work-item (gymrs_set_tuning; HIP launches only) against 4, 2^20 CartPole lanes" >> $L
timeout 600 python tools/step_timer.py --env 0 --n $((1<<20)) --steps 16000 --reps 5 --aql 0 --vec 4 --nbuf 32 2>&1 | grep -v "^ring\|amdgpu.ids" >> $L
timeout 600 python tools/step_timer.py --env 0 --n $((1<<20)) --steps 16000 --reps 5 --aql 0 --vec 8 --nbuf 32 2>&1 | grep -v "^ring\|amdgpu.ids" >> $L
cat $L
timeout 900 python -m pytest tests/test_gpu_bench_contract.py -x -q -k "in_process" 2>&1 | tail -6 | tee -a gpurun_out/r05/pytest_call06.log
python bench.py --in-process --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/r05/bench_in_process_1.json 2> gpurun_out/r05/bench_in_process_1.err; cat gpurun_out/r05/bench_in_process_1.json
