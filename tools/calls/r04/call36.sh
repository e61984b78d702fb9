#!/bin/bash
# round 4, call 36: the round-end sequence as the driver runs it, on the final tree: smoke(), pytest -m gpu, the bench command (timed)
set -u
OUT=gpurun_out/r04_c36; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import bench; print('kernel_source_sha16', bench.kernel_source_sha16())" > $OUT/status.log 2>&1
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc $? $(tail -1 $OUT/smoke.log)" >> $OUT/status.log
timeout 1800 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $? $(grep -E 'passed|failed' $OUT/pytest_gpu.log | tail -1)" >> $OUT/status.log
t0=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $? in $(( $(date +%s) - t0 )) s" >> $OUT/status.log
t0=$(date +%s); timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench (no flags) rc $? in $(( $(date +%s) - t0 )) s" >> $OUT/status.log
echo done >> $OUT/status.log
