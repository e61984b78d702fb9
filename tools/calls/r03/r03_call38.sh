#!/bin/bash
# the GPU suite on the alternative submission paths (HIP launches only; synchronous hand-over), then 8 and 2 ranks sharing the one GPU through the
# driver's own launcher form (gloo stand-in is in the tests; here the real RCCL path)
set -u
OUT=gpurun_out/r03_c38; mkdir -p $OUT
GYMRS_AQL=0 timeout 900 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_aql0.log 2>&1; echo "AQL=0 rc=$? $(grep -E 'passed|failed' $OUT/pytest_aql0.log | tail -1)"
GYMRS_AQL_SYNC=1 timeout 900 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_sync.log 2>&1; echo "SYNC=1 rc=$? $(grep -E 'passed|failed' $OUT/pytest_sync.log | tail -1)"
for n in 2 8; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n bench.py --gpus $n --steps 20 --warmup 5 --oversubscribe > $OUT/bench_n$n.json 2> $OUT/bench_n$n.err
  echo "n=$n rc=$? $(tail -1 $OUT/bench_n$n.json | cut -c1-240)"
done
