#!/bin/bash
set -u
OUT=gpurun_out/r03_c15
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $OUT/pytest_all.log
cat $OUT/pytest_all.log
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 2 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err
timeout 600 python bench.py --cpu-seconds 0 --no-probe --no-configs > $OUT/bench_default.json 2> $OUT/bench_default.err
GYMRS_AQL=0 timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-probe --no-configs > $OUT/bench_driver_form_hip.json 2> $OUT/bench_driver_form_hip.err
python - <<'PY'
import json
for f in ("bench_driver_form","bench_default","bench_driver_form_hip"):
    try: d=json.load(open(f"gpurun_out/r03_c15/{f}.json"))
    except Exception as e:
        print(f,"ERR",e); print(open(f"gpurun_out/r03_c15/{f}.err").read()[-1500:]); continue
    t=d["timing"]; print(f,"value %.4e launch_us %.3f frac %.3f"%(d["value"],d["roofline"]["launch_us"],d["roofline"]["frac"]), t["event_us_per_step"], d["config"]["submission"][:20])
    for k,c in (d.get("configs") or {}).items(): print("   ",k,"%.4e"%c["value"],round(c["launch_us"],3),round(c["frac"],3),c.get("frac_of_same_footprint_copy"))
PY
echo done15
