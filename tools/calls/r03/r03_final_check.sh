#!/bin/bash
# what the driver does at round end, in its order: the GPU suite, smoke(), the bench line
set -u
OUT=gpurun_out/r03_final
mkdir -p $OUT
python -m pytest tests/ -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$? $(grep -E 'passed|failed' $OUT/pytest.log | tail -1)"
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $OUT/smoke.log | cut -c1-160)"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -1 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_final/bench.json"))
print("value %.4e ms_per_step %.6f launch_us %.3f frac %.3f spread %.4f"%(d["value"],d["ms_per_step"],d["roofline"]["launch_us"],d["roofline"]["frac"],d["timing"]["event_us_per_step"]["spread"]), d["roofline"]["kernel"], d["roofline"].get("traffic"), d["roofline"].get("traffic_source"))
print(d["config"]["submission"][:60]); print({k:round(v["value"]/1e11,3) for k,v in d["configs"].items()}); print("cpu", d["cpu_baseline"]["value"])
PY
