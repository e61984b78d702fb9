#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the parts of the round-3 evidence set that carry the kernel-source hash, re-collected after the last source change -> gpurun_out/<tag>_*  (then tools/publish_evidence.sh <tag> r03).
#   gpurun --timeout 2400 -- 'bash tools/collect_evidence_r03.sh r03a'
# PMC passes are separate runs without any trace domain besides the counters (gpurun refuses mixes).
set -u
TAG=${1:-r03}
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
PIN="taskset -c 4-7"
SHA=$(python -c "import bench; print(bench.kernel_source_sha16())")
echo "kernel_source_sha16 $SHA" > "$OUT/${TAG}_sha.txt"

# 0. the GPU suite
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/${TAG}_pytest_gpu_full.log" 2>&1; grep -E "passed|failed" "$OUT/${TAG}_pytest_gpu_full.log" | tail -2 > "$OUT/${TAG}_pytest_gpu.log"

# 1. HBM traffic (PMC FETCH_SIZE / WRITE_SIZE, one counter per pass) of the kernels as the bench runs them (chains)
for env in cartpole mountain_car pendulum; do
    python bench.py --env $env --pmc-traffic --cpu-seconds 0 --no-probe > "$OUT/${TAG}_bench_pmc_${env}.json" 2> "$OUT/${TAG}_bench_pmc_${env}.err"
done
cp profiles/pmc_traffic.json "$OUT/${TAG}_pmc_traffic.json"

# 2. the bench lines: the driver's own command, the default form per env, and the same through HIP launches only
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/${TAG}_bench_driver_form.json" 2> "$OUT/${TAG}_bench_driver_form.err"
for env in cartpole mountain_car pendulum; do
    python bench.py --env $env > "$OUT/${TAG}_bench_${env}.json" 2> "$OUT/${TAG}_bench_${env}.err"
    GYMRS_AQL=0 python bench.py --env $env --cpu-seconds 0 > "$OUT/${TAG}_bench_${env}_hip_launches.json" 2> "$OUT/${TAG}_bench_${env}_hip_launches.err"
done
GYMRS_AQL=0 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 > "$OUT/${TAG}_bench_driver_form_hip_launches.json" 2>/dev/null

# 3. kernel trace of the bench command: chains, and HIP launches; the step kernel's (start, end) rows are kept as CSV
cd /tmp
for mode in chain hip; do
    rm -rf "$OUT/${TAG}_kt_$mode"
    GYMRS_AQL=$([ $mode = chain ] && echo 1 || echo 0) rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_kt_$mode" -o r -- python "$REPO/bench.py" --steps 1000 --warmup 200 \
        --cpu-seconds 0 --no-probe --no-configs > "$OUT/${TAG}_bench_cartpole_under_rocprof_$mode.json" 2> "$OUT/${TAG}_kt_$mode.err"
    DB=$(find $OUT/${TAG}_kt_$mode -name '*_results.db' | head -1)
    python "$REPO/tools/summarize_rocprof.py" kernel "$DB" "$OUT/${TAG}_kernel_trace_stats_cartpole_$mode.txt" > /dev/null
    python - "$DB" "$OUT/${TAG}_kernel_trace_cartpole_$mode.csv.gz" <<'PY'
import gzip, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
with gzip.open(sys.argv[2], "wt") as f:
    f.write("name,start_ns,end_ns\n")
    for n, s, e in rows:
        f.write(f"{n.split('(')[0][:60]},{s},{e}\n")
PY
    rm -rf "$OUT/${TAG}_kt_$mode"
done

# 6. fused rollout: SQ_INSTS_VALU per launch (the numerator of its VALU-issue roofline) and the bench lines
for env in cartpole mountain_car pendulum; do
    for rec in "" "--record"; do
        key=${env}${rec:+_recorded}
        rm -rf "$OUT/${TAG}_pmc_valu_${key}"
        rocprofv3 --pmc SQ_INSTS_VALU -d "$OUT/${TAG}_pmc_valu_${key}" -o r -- python "$REPO/bench.py" --env $env --rollout 128 $rec --steps 256 --warmup 128 \
            --cpu-seconds 0 --repetitions 2 > /dev/null 2> "$OUT/${TAG}_pmc_valu_${key}.err"
    done
done
cd "$REPO"
for env in cartpole mountain_car pendulum; do
    n=$([ $env = pendulum ] && echo 4194304 || echo 1048576)
    python tools/summarize_rocprof.py valu "$(find $OUT/${TAG}_pmc_valu_${env} -name '*_results.db' | head -1)" $env $n 128 $SHA profiles/pmc_valu.json > /dev/null
    python tools/summarize_rocprof.py valu "$(find $OUT/${TAG}_pmc_valu_${env}_recorded -name '*_results.db' | head -1)" ${env}_recorded $n 128 $SHA profiles/pmc_valu.json > /dev/null
done
cp profiles/pmc_valu.json "$OUT/${TAG}_pmc_valu.json"
rm -rf $OUT/${TAG}_pmc_valu_*/
for env in cartpole mountain_car pendulum; do
    python bench.py --env $env --rollout 128 --steps 2048 --warmup 256 --cpu-seconds 0 > "$OUT/${TAG}_bench_rollout_${env}.json" 2>/dev/null
    python bench.py --env $env --rollout 128 --record --steps 1024 --warmup 128 --cpu-seconds 0 > "$OUT/${TAG}_bench_rollout_recorded_${env}.json" 2>/dev/null
done

# 8. the slow-mode sample of this box
SLOW_MODE_DETAIL="$OUT/${TAG}_slow_mode.detail.json" timeout 120 python tools/exp_slow_mode.py --seconds 5 --smi-ms 5 --tag $TAG > "$OUT/${TAG}_slow_mode.jsonl" 2> /dev/null
rm -f "$OUT/${TAG}_slow_mode.detail.json"
echo refresh-done
