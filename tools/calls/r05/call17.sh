#!/bin/bash
# call 17: the one wrong answer of soak part 3 (test_eight_ranks_share_the_gpu[32768]) did not reproduce in 170 stand-alone runs of its command; here it runs the way the
# suite runs it -- from a pytest process that itself holds engines and parked chain queues on the GPU (the chain module first) -- N times; rank records now carry each
# rank's own statistics, so a failure says which block is off
bash tools/calls/r05/soak.sh eight_ranks_in_suite_context 60 X=1 -- tests/test_gpu_aql_chain.py tests/test_gpu_bench_contract.py -k "three_engines or every_flag_set or (eight_ranks and 32768)"
