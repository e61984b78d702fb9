#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): collects everything profiles/ is built from into gpurun_out/<tag>_*.
#   gpurun --timeout 1500 -- 'bash tools/collect_evidence.sh r01f'
# Then, back in the authoring container: python tools/summarize_rocprof.py ... (see profiles/README.md).
# PMC passes are separate runs without any trace domain besides the counters (gpurun refuses mixes).
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD

for env in cartpole mountain_car pendulum; do
    python bench.py --env $env > "$OUT/${TAG}_bench_${env}.json" 2> "$OUT/${TAG}_bench_${env}.err"
done
for env in cartpole mountain_car pendulum; do
    python bench.py --env $env --rollout 128 --steps 2048 --warmup 256 --cpu-seconds 0 > "$OUT/${TAG}_bench_rollout_${env}.json" 2>/dev/null
done
for env in cartpole mountain_car pendulum; do
    python bench.py --env $env --rollout 128 --record --steps 1024 --warmup 128 --cpu-seconds 0 > "$OUT/${TAG}_bench_rollout_recorded_${env}.json" 2>/dev/null
done
for n in 1024 16384 131072; do
    for g in "" "--graph"; do
        python bench.py --n-envs $n --steps 4000 --warmup 400 --cpu-seconds 0 $g 2>/dev/null
    done
done > "$OUT/${TAG}_small_batch_graph.jsonl"

# kernel trace (+ the bench line printed under the tracer)
cd /tmp
rm -rf "$OUT/${TAG}_kt"
rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_kt" -o r -- python "$REPO/bench.py" --steps 2000 --warmup 200 --cpu-seconds 0 \
    > "$OUT/${TAG}_kt_bench.json" 2> "$OUT/${TAG}_kt.err"
# HBM traffic counters, one counter per pass
for env in cartpole mountain_car pendulum; do
    for ctr in FETCH_SIZE WRITE_SIZE; do
        d="$OUT/${TAG}_pmc_${ctr}_${env}"
        rm -rf "$d"
        rocprofv3 --pmc $ctr -d "$d" -o r -- python "$REPO/bench.py" --env $env --steps 300 --warmup 100 --cpu-seconds 0 \
            > /dev/null 2> "$d.err"
    done
done
cd "$REPO"
find "$OUT" -name "*.db" -newer tools/collect_evidence.sh | head -20
echo evidence-done
