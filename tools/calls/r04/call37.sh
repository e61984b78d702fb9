#!/bin/bash
# round 4, call 37: why does the driver's form (--steps 20 --warmup 5: 780-800 launches per call) settle at ~6.50 us per launch and the default form
# (--steps 1000 --warmup 200: 1000 per call) at ~6.37 on the same box?  K, W and the launches per call varied, two rounds
set -u
OUT=gpurun_out/r04_c37; mkdir -p $OUT
export TMPDIR=/tmp
for round in 1 2; do
  for cfg in "20 5" "25 5" "40 5" "1000 5" "20 200" "1000 200" "500 200" "16 5" "32 5"; do
    set -- $cfg
    timeout 300 python bench.py --gpus 1 --steps $1 --warmup $2 --cpu-seconds 0 --no-configs --no-probe > $OUT/b_${1}_${2}_r$round.json 2> $OUT/b_${1}_${2}_r$round.err
    python - $OUT/b_${1}_${2}_r$round.json "$1 $2" >> $OUT/status.log <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); t = d["timing"]; n = t["steps_per_repetition"]
v = d["paths"]["per_step_visible"]
print("K W =", sys.argv[2], "| launches per call", n, "calibration", t["calibration_calls"], "| visible us per launch per repetition", [round(x * 1e3 / n, 3) for x in v["event_ms_per_repetition"]],
      "| host enqueue", round(v["host_enqueue_us_per_step"]["median"], 2))
PY
  done
done
echo done >> $OUT/status.log
