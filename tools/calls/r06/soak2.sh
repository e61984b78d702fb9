#!/bin/bash
# round 6, GPU call 15: the FINAL tree (CartPole's loads ahead of the argument fetch, 862e341a5d69b8b3): the GPU suite four times as the driver runs it, smoke(), then
# the hand-over audit and the cold-start amplifier again (the chain kernels are the same body: they changed too)
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/calls/r06/soak.sh 4 b
OUT=gpurun_out/r06_audit2.log
: > $OUT
run() { echo "# $*" >> $OUT; timeout 900 python tools/handover_amp.py "$@" 2>&1 | grep -v "^/usr/lib/python3.10/multiprocessing\|^  warnings.warn" >> $OUT; echo "# rc=${PIPESTATUS[0]}" >> $OUT; }
run --procs 8 --seconds 120 --mode both  --audit --flags 7 --handover kernel
run --procs 8 --seconds 90  --mode chain --audit --flags 3 --handover auto
run --procs 8 --seconds 60  --mode hip   --audit --flags 7
echo "# coldstart_amp --seconds 300" >> $OUT
timeout 900 python tools/coldstart_amp.py --seconds 300 2>&1 | grep -v "resource_tracker\|warnings.warn" >> $OUT
python - <<'PY'
import json
for l in open('gpurun_out/r06_audit2.log'):
    if l.startswith('{'):
        r=json.loads(l); print({k:r.get(k) for k in ('tool','mode','flags','iterations','chain_calls','wrong_iterations','loud_failures','rounds','process_runs','wrong','errors')})
PY
