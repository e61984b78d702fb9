#!/bin/bash
# round 4, call 33: the final library (with the reward-store elision): the GPU suite on the three alternative submission paths, three fuzz seeds and the
# big-lane fuzz cases
set -u
OUT=gpurun_out/r04_c33; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import bench; print('kernel_source_sha16', bench.kernel_source_sha16())" > $OUT/status.log 2>&1
GYMRS_AQL=0 timeout 900 python -m pytest tests -m "gpu and not perf" -x -q -p no:cacheprovider > $OUT/suite_aql0.log 2>&1; echo "GYMRS_AQL=0 rc $? $(grep -E 'passed|failed' $OUT/suite_aql0.log | tail -1)" >> $OUT/status.log
GYMRS_AQL_SYNC=1 timeout 900 python -m pytest tests -m "gpu and not perf" -x -q -p no:cacheprovider > $OUT/suite_sync.log 2>&1; echo "GYMRS_AQL_SYNC=1 rc $? $(grep -E 'passed|failed' $OUT/suite_sync.log | tail -1)" >> $OUT/status.log
GYMRS_AQL_HANDOVER=sync timeout 900 python -m pytest tests -m "gpu and not perf" -x -q -p no:cacheprovider > $OUT/suite_handover_sync.log 2>&1; echo "GYMRS_AQL_HANDOVER=sync rc $? $(grep -E 'passed|failed' $OUT/suite_handover_sync.log | tail -1)" >> $OUT/status.log
for seed in 81 82; do
  timeout 900 python tools/fuzz_engine_vs_twin.py --cases 220 --seed $seed > $OUT/fuzz_seed$seed.log 2>&1; echo "fuzz seed $seed rc $? $(tail -1 $OUT/fuzz_seed$seed.log)" >> $OUT/status.log
done
timeout 900 python tools/fuzz_engine_vs_twin.py --cases 6 --seed 84 --min-lanes 2500000 --max-lanes 5000000 --ops 10 > $OUT/fuzz_big.log 2>&1; echo "fuzz big rc $? $(tail -1 $OUT/fuzz_big.log)" >> $OUT/status.log
echo done >> $OUT/status.log
