#!/bin/bash
# round 4, call 34: the chain soak at 2^22 lanes (CartPole's reward-store elision active), final library
set -u
OUT=gpurun_out/r04_c34; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python tests/soak_chains.py 400000 22 > $OUT/soak_chains_2p22.log 2>&1; echo "soak rc $?" >> $OUT/status.log
echo done >> $OUT/status.log
