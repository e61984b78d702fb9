#!/bin/bash
# round 4, call 18: random operation sequences against the CPU f32 twin with this round's library (per-chain XCD check, hint variants incl. the new
# outputs-only instantiation of the HIP launch table, 256 / 512 work-item groups by size): four seeds, and six cases at 1.4-5 M lanes
set -u
OUT=gpurun_out/r04_c18; mkdir -p $OUT
export TMPDIR=/tmp
for seed in 41 42 43 44; do
  timeout 900 python tools/fuzz_engine_vs_twin.py --cases 220 --seed $seed > $OUT/fuzz_seed$seed.log 2>&1; echo "seed $seed rc $?" >> $OUT/status.log
done
timeout 900 python tools/fuzz_engine_vs_twin.py --cases 6 --seed 45 --min-lanes 1400000 --max-lanes 5000000 --ops 10 > $OUT/fuzz_big.log 2>&1; echo "big rc $?" >> $OUT/status.log
echo done >> $OUT/status.log
