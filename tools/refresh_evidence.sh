#!/bin/bash
# Runs ON THE GPU BOX (through gpurun), ONCE, from the final tree of the round (VERDICT r3 "next" #9): everything under profiles/ that carries the
# kernel-source hash or describes the final library -> gpurun_out/<tag>_*; tools/publish_evidence.sh <tag> <round> then copies it into profiles/.
#   gpurun --timeout 3000 -- 'bash tools/refresh_evidence.sh r04'
# Counter passes are separate runs without any trace domain besides the counters (gpurun refuses mixes).
set -u
TAG=${1:-r05}
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
SHA=$(python -c "import bench; print(bench.kernel_source_sha16())")
echo "kernel_source_sha16 $SHA" > "$OUT/${TAG}_sha.txt"
# the counter tool of step 1 is git-ignored like every built file: a rebuilt container does not have it (round 6 lost a pass to that)
[ -f tools/devcount/libgymrs_devcount.so ] || g++ -O2 -std=c++17 -fPIC -shared -D__HIP_PLATFORM_AMD__ tools/devcount/gymrs_devcount.cpp -I/opt/rocm/include -L/opt/rocm/lib -lrocprofiler-sdk -o tools/devcount/libgymrs_devcount.so

# 0. the GPU suite and smoke(), as the driver runs them
timeout 1800 python -m pytest tests -m gpu -x -q > "$OUT/${TAG}_pytest_gpu_full.log" 2>&1; grep -E "passed|failed" "$OUT/${TAG}_pytest_gpu_full.log" | tail -2 > "$OUT/${TAG}_pytest_gpu.log"
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/${TAG}_smoke.log" 2>&1

# 1. fabric traffic of FREE-RUNNING launches, per config and call shape (device-wide counters); the bench lines below read it from profiles/
timeout 1200 python tools/devcount/collect.py --out "$OUT/${TAG}_devcount_traffic.json" > "$OUT/${TAG}_devcount_collect.log" 2>&1
cp "$OUT/${TAG}_devcount_traffic.json" profiles/devcount_traffic.json

# 2. per-dispatch PMC (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass) of the same kernels: each launch in isolation
for env in cartpole mountain_car pendulum; do
    timeout 900 python bench.py --env $env --pmc-traffic --cpu-seconds 0 --no-probe --no-configs --full-out "$OUT/${TAG}_bench_pmc_${env}_full.json" > "$OUT/${TAG}_bench_pmc_${env}.json" 2> "$OUT/${TAG}_bench_pmc_${env}.err"
done
cp profiles/pmc_traffic.json "$OUT/${TAG}_pmc_traffic.json"

# 3. the bench lines: the driver's own command, the default form per env
# (since round 5 stdout carries the <= 4 KB digest the driver keeps; the complete record goes to --full-out)
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --full-out "$OUT/${TAG}_bench_driver_form_full.json" > "$OUT/${TAG}_bench_driver_form.json" 2> "$OUT/${TAG}_bench_driver_form.err"
for env in cartpole mountain_car pendulum; do
    timeout 900 python bench.py --env $env --full-out "$OUT/${TAG}_bench_${env}_full.json" > "$OUT/${TAG}_bench_${env}.json" 2> "$OUT/${TAG}_bench_${env}.err"
done
# 3b. the same workload through the C ABI's native in-process sharder (one block on this box's one GPU; 4 blocks sharing it: a TEST of the path, not a rate)
timeout 600 python bench.py --in-process --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --full-out "$OUT/${TAG}_bench_in_process_full.json" > "$OUT/${TAG}_bench_in_process.json" 2> "$OUT/${TAG}_bench_in_process.err"
timeout 600 python bench.py --in-process --gpus 4 --oversubscribe --n-envs 262144 --steps 20 --warmup 5 --cpu-seconds 0 --full-out "$OUT/${TAG}_bench_in_process_4_blocks_one_gpu_full.json" > "$OUT/${TAG}_bench_in_process_4_blocks_one_gpu.json" 2> "$OUT/${TAG}_bench_in_process_4.err"

# 4. kernel trace of the bench command (both call shapes run in it); the step kernels' (start, end) rows are kept as CSV
cd /tmp
rm -rf "$OUT/${TAG}_kt"
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_kt" -o r -- python "$REPO/bench.py" --steps 1000 --warmup 200 --cpu-seconds 0 --no-probe --no-configs --min-repetition-ms 5 --full-out "$OUT/${TAG}_bench_cartpole_under_rocprof_full.json" \
    > "$OUT/${TAG}_bench_cartpole_under_rocprof.json" 2> "$OUT/${TAG}_kt.err"
DB=$(find $OUT/${TAG}_kt -name '*_results.db' | head -1)
python "$REPO/tools/summarize_rocprof.py" kernel "$DB" "$OUT/${TAG}_kernel_trace_stats_cartpole.txt" > /dev/null
python - "$DB" "$OUT/${TAG}_kernel_trace_cartpole.csv.gz" "$OUT/${TAG}_kernel_trace_by_call_shape.txt" <<'PY'
import gzip, sqlite3, statistics, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
with gzip.open(sys.argv[2], "wt") as f:
    f.write("name,start_ns,end_ns\n")
    for n, s, e in rows:
        f.write(f"{n.split('(')[0][:60]},{s},{e}\n")
with open(sys.argv[3], "w") as f:
    for label, pred in (("per_step_visible (HIP launches)", lambda n: "step_kernel" in n), ("chain", lambda n: n.startswith("gymrs_aql_cartpole") and not n.startswith("gymrs_aql_cartpole_f3_t512_nt")),
                        ("per_step_visible through the engine's queue (GYMRS_AQL=2)", lambda n: n.startswith("gymrs_aql_cartpole_f3_t512_nt"))):
        ks = [(s, e) for n, s, e in rows if pred(n)]
        if len(ks) < 50:
            continue
        ks = ks[len(ks) // 10:]
        durs = [e - s for s, e in ks]
        gaps = [ks[i + 1][0] - ks[i][0] for i in range(len(ks) - 1) if ks[i + 1][0] - ks[i][0] < 50000]
        f.write(f"{label}: n={len(ks)} duration median {statistics.median(durs):.0f} ns mean {statistics.mean(durs):.0f} ns; start-to-start median {statistics.median(gaps):.0f} ns mean {statistics.mean(gaps):.0f} ns\n")
PY
rm -rf "$OUT/${TAG}_kt"

# 5. fused rollout: SQ_INSTS_VALU per launch (the numerator of its VALU-issue roofline) and the bench lines
for env in cartpole mountain_car pendulum; do
    for rec in "" "--record"; do
        key=${env}${rec:+_recorded}
        rm -rf "$OUT/${TAG}_pmc_valu_${key}"
        timeout 600 rocprofv3 --pmc SQ_INSTS_VALU -d "$OUT/${TAG}_pmc_valu_${key}" -o r -- python "$REPO/bench.py" --env $env --rollout 128 $rec --steps 256 --warmup 128 \
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
    timeout 600 python bench.py --env $env --rollout 128 --steps 2048 --warmup 256 --cpu-seconds 0 > "$OUT/${TAG}_bench_rollout_${env}.json" 2>/dev/null
    timeout 600 python bench.py --env $env --rollout 128 --record --steps 1024 --warmup 128 --cpu-seconds 0 > "$OUT/${TAG}_bench_rollout_recorded_${env}.json" 2>/dev/null
done

# 6. step vs in-place copy over sizes, both call shapes
for env in 0 1 2; do
    sizes=$([ $env = 2 ] && echo 18,20,21,22,23,24 || echo 18,20,21,22,23,24,25)
    for aql in 1 0; do
        GYMRS_AQL=$aql timeout 900 python tools/size_sweep.py --env $env --sizes $sizes >> "$OUT/${TAG}_size_sweep_env${env}_aql$aql.log" 2>&1
    done
done
# 7. the per-step-visible shape through HIP launches / chains / the engine's queue with HIP's header, by env and size (VERDICT r4 "next" #4)
L="$OUT/${TAG}_submission_by_size.log"
: > "$L"
for env in 0 1 2; do for lg in 20 21 22; do
    echo "# env $env 2^$lg lanes, 8 action buffers" >> "$L"
    timeout 600 python tools/step_timer.py --env $env --n $((1<<lg)) --steps $((lg == 20 ? 16000 : 6000)) --reps 5 --aql 0,1,2 --nbuf 8 2>&1 | grep "us median" >> "$L"
done; done
echo refresh-done
