#!/bin/bash
# round 6, GPU call 7: steps 1 and 3 of tools/refresh_evidence.sh again -- call 6 ran without tools/devcount/libgymrs_devcount.so (a rebuilt container: the tool library is
# git-ignored and nothing rebuilt it; refresh_evidence.sh now builds it first), so its bench lines carry no `traffic` -- then the packed-math experiment (pk1.sh)
set -u
TAG=r06
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1200 python tools/devcount/collect.py --out "$OUT/${TAG}_devcount_traffic.json" > "$OUT/${TAG}_devcount_collect.log" 2>&1
cp "$OUT/${TAG}_devcount_traffic.json" profiles/devcount_traffic.json
cp "$OUT/${TAG}_pmc_traffic.json" profiles/pmc_traffic.json 2>/dev/null
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --full-out "$OUT/${TAG}_bench_driver_form_full.json" > "$OUT/${TAG}_bench_driver_form.json" 2> "$OUT/${TAG}_bench_driver_form.err"
for env in cartpole mountain_car pendulum; do
    timeout 900 python bench.py --env $env --full-out "$OUT/${TAG}_bench_${env}_full.json" > "$OUT/${TAG}_bench_${env}.json" 2> "$OUT/${TAG}_bench_${env}.err"
done
timeout 600 python bench.py --in-process --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --full-out "$OUT/${TAG}_bench_in_process_full.json" > "$OUT/${TAG}_bench_in_process.json" 2> "$OUT/${TAG}_bench_in_process.err"
timeout 600 python bench.py --in-process --gpus 4 --oversubscribe --n-envs 262144 --steps 20 --warmup 5 --cpu-seconds 0 --full-out "$OUT/${TAG}_bench_in_process_4_blocks_one_gpu_full.json" > "$OUT/${TAG}_bench_in_process_4_blocks_one_gpu.json" 2> "$OUT/${TAG}_bench_in_process_4.err"
tail -3 "$OUT/${TAG}_devcount_collect.log"
cat "$OUT/${TAG}_bench_driver_form.json"
bash tools/calls/r06/pk1.sh
