#!/bin/bash
# round 6, last GPU call: what the driver runs at round end, in its order, on a fresh box from the committed tree: the GPU suite (-x), smoke(), the bench command
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06_final_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r06_final_pytest.log
grep -E "passed|failed|rc=" gpurun_out/r06_final_pytest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_final_bench.json 2> gpurun_out/r06_final_bench.err; echo "bench rc=$?"
cat gpurun_out/r06_final_bench.json | cut -c1-1500
