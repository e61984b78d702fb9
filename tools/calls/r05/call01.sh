#!/bin/bash
# round 5, call 1: does the driver's command print a line the driver can keep?  + the bench contract tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/bench_driver_form.json 2> gpurun_out/r05/bench_driver_form.err
echo "rc $? bytes $(wc -c < gpurun_out/r05/bench_driver_form.json)"
cp bench_full.json gpurun_out/r05/bench_driver_form_full.json
cat gpurun_out/r05/bench_driver_form.json
timeout 1500 python -m pytest tests/test_gpu_bench_contract.py -x -q 2>&1 | tail -15 | tee gpurun_out/r05/pytest_bench_contract.log
