#!/bin/bash
# round 6: the GPU suite on the tree with read-only / baseline statistics and HIP launches as gymrs_step_many's default
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r06_pytest_gpu.log 2>&1
echo "rc=$?" >> gpurun_out/r06_pytest_gpu.log
grep -n "^FAILED\|passed\|failed" gpurun_out/r06_pytest_gpu.log | tail -20
timeout 300 python __graft_entry__.py smoke > gpurun_out/r06_smoke.log 2>&1; tail -3 gpurun_out/r06_smoke.log
