#!/bin/bash
# round 5, call 2: the native sharder + the copy tool on hardware, then the whole GPU suite
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_sharded_native.py tests/test_gpu_params_and_serde.py tests/test_gpu_envs.py -x -q 2>&1 | tail -25 | tee gpurun_out/r05/pytest_sharded.log
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r05/pytest_gpu_call02.log
