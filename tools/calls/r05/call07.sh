#!/bin/bash
# call 7: soak, part 1 -- the product-heavy part of the suite (everything but the bench-contract module and the perf-marked tests), 40 x with chains, 40 x with GYMRS_AQL=0
bash tools/calls/r05/soak.sh chains 40 X=1 -- tests -m "gpu and not perf" --ignore=tests/test_gpu_bench_contract.py
bash tools/calls/r05/soak.sh hip_launches 40 GYMRS_AQL=0 -- tests -m "gpu and not perf" --ignore=tests/test_gpu_bench_contract.py
