#!/bin/bash
# call 11: soak, part 2 (after the tests' stream races were fixed): the product-heavy part 30 x with chains, then the FULL suite (bench contract and perf tests included) 8 x
bash tools/calls/r05/soak.sh chains_after_fix 30 X=1 -- tests -m "gpu and not perf" --ignore=tests/test_gpu_bench_contract.py
bash tools/calls/r05/soak.sh full_suite 8 X=1 -- tests -m gpu
