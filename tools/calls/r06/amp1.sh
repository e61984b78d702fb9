#!/bin/bash
# round 6, GPU call 1: the hand-over amplifier's A/B matrix (tools/handover_amp.py) -- where does the wrong episode count come from, and how often?
mkdir -p gpurun_out
OUT=gpurun_out/r06_amp1.log
: > $OUT
python -c "import torch; print(torch.cuda.get_device_name(0))" >> $OUT 2>&1
run() { echo "# $*" >> $OUT; timeout 900 python tools/handover_amp.py "$@" >> $OUT 2>&1; echo "# rc=$?" >> $OUT; }
run --procs 8 --seconds 60 --mode both --handover kernel --lockstep
run --procs 8 --seconds 45 --mode hip --lockstep
run --procs 8 --seconds 45 --mode chain --handover kernel --lockstep
run --procs 8 --seconds 45 --mode both --handover sync --lockstep
run --procs 8 --seconds 45 --mode both --handover kernel --lockstep --check-clear
run --procs 8 --seconds 45 --mode both --handover kernel
tail -c 6000 $OUT
