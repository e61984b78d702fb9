#!/bin/bash
# round 4, call 38: us per launch of the per-step-visible shape against the number of launches per gymrs_step_many call (= per timed repetition)
set -u
OUT=gpurun_out/r04_c38; mkdir -p $OUT
export TMPDIR=/tmp
for round in 1 2; do
  for k in 800 850 900 950 1000 1200 1600 3200 10000 400 800; do
    timeout 300 python bench.py --gpus 1 --steps $k --warmup 5 --cpu-seconds 0 --no-configs --no-probe > $OUT/b_${k}_r$round.json 2> $OUT/b_${k}_r$round.err
    python - $OUT/b_${k}_r$round.json "$k" >> $OUT/status.log <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); t = d["timing"]; n = t["steps_per_repetition"]
v = d["paths"]["per_step_visible"]; c = d["paths"]["chain"]
print("K =", sys.argv[2], "| launches per call", n, "| visible us per launch per repetition", [round(x * 1e3 / n, 3) for x in v["event_ms_per_repetition"]],
      "| host enqueue", round(v["host_enqueue_us_per_step"]["median"], 2), "| chain median", round(c["launch_us"], 3))
PY
  done
done
echo done >> $OUT/status.log
