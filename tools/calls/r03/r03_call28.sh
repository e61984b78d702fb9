#!/bin/bash
set -u
OUT=gpurun_out
TAG=r03d
export TMPDIR=/tmp
for env in cartpole mountain_car pendulum; do
    python bench.py --env $env --pmc-traffic --cpu-seconds 0 --no-probe --no-configs > "$OUT/${TAG}_bench_pmc_${env}.json" 2> "$OUT/${TAG}_bench_pmc_${env}.err"
    echo "$env rc=$?"
done
cp profiles/pmc_traffic.json "$OUT/${TAG}_pmc_traffic.json"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03d_pmc_traffic.json")); print(d["kernel_source_sha16"])
for e in ("cartpole","mountain_car","pendulum"):
    r=d[e]; print(e, "hip %.1f MB (fetch %.1f write %.1f)"%(r["bytes_per_launch"]/1e6,r["fetch_bytes"]/1e6,r["write_bytes"]/1e6), "chain", {k:round(v/1e6,1) for k,v in r.get("chain",{}).items() if k.endswith("bytes") or k=="bytes_per_launch"})
PY
