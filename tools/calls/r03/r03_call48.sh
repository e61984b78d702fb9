#!/bin/bash
# the launcher form with one rank (the communicator path with world_size 1: gloo control plane + the C ABI's RCCL all-reduce), the same with the
# control plane on torch's nccl backend, then the bench-contract tests
set -u
OUT=gpurun_out/r03_c48; mkdir -p $OUT
for v in 0 1; do
  GYMRS_BENCH_FORCE_DIST=1 GYMRS_BENCH_NCCL=$v timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2961$v bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 1 > $OUT/bench_nccl$v.json 2> $OUT/bench_nccl$v.err
  echo "nccl=$v rc=$?"; python - <<PY
import json
d=json.loads(open("$OUT/bench_nccl$v.json").read().strip().splitlines()[-1])
print("  value %.4e"%d["value"], d["config"]["stats_allreduce"], "|", d["config"]["control_plane"], "|", d["config"].get("comm_watchdog"), "|", d["config"].get("stats_allreduce_note"))
PY
done
timeout 900 python -m pytest tests/test_gpu_bench_contract.py -q -m gpu > $OUT/contract.log 2>&1; tail -2 $OUT/contract.log
