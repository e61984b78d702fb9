#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): collects everything profiles/ is built from into gpurun_out/<tag>_*.
#   gpurun --timeout 2400 -- 'bash tools/collect_evidence.sh r02a'
# Then, back in the authoring container:  bash tools/summarize_evidence.sh r02a   (text/JSON summaries -> profiles/)
# PMC passes are separate runs without any trace domain besides the counters (gpurun refuses mixes).
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
# bench.py confines itself to 4 CPUs (sharded.pin_rank_to_cpus); the developer tools get the same treatment from outside
PIN="taskset -c 4-7"
SHA=$(python -c "import bench; print(bench.kernel_source_sha16())")
echo "kernel_source_sha16 $SHA" > "$OUT/${TAG}_sha.txt"

# 1. HBM traffic (PMC FETCH_SIZE / WRITE_SIZE, one counter per pass): bench.py runs the two rocprofv3 passes itself and
#    refreshes profiles/pmc_traffic.json (stamped with the kernel source hash), which is copied back below.
#    FIRST, so that the bench lines of step 2 carry roofline.traffic of these very kernels
for env in cartpole mountain_car pendulum; do
    python bench.py --env $env --pmc-traffic --cpu-seconds 0 --no-probe > "$OUT/${TAG}_bench_pmc_${env}.json" 2> "$OUT/${TAG}_bench_pmc_${env}.err"
done
cp profiles/pmc_traffic.json "$OUT/${TAG}_pmc_traffic.json"

# 2. the bench lines: the driver's own command, the default form, the other configs
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/${TAG}_bench_driver_form.json" 2> "$OUT/${TAG}_bench_driver_form.err"
for env in cartpole mountain_car pendulum; do
    python bench.py --env $env > "$OUT/${TAG}_bench_${env}.json" 2> "$OUT/${TAG}_bench_${env}.err"
done
python bench.py --cpu-seconds 0 --no-probe --graph > "$OUT/${TAG}_bench_cartpole_graph.json" 2>/dev/null

# 3. kernel trace of the bench command, eager and as graph replays (under the tracer eager launches are host-bound)
cd /tmp
for mode in "" "--graph"; do
    suffix=${mode:+_graph}
    rm -rf "$OUT/${TAG}_kt${suffix}"
    rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_kt${suffix}" -o r -- python "$REPO/bench.py" --steps 1000 --warmup 200 --cpu-seconds 0 --no-probe $mode \
        > "$OUT/${TAG}_kt${suffix}_bench.json" 2> "$OUT/${TAG}_kt${suffix}.err"
    python "$REPO/tools/summarize_rocprof.py" kernel "$(find $OUT/${TAG}_kt${suffix} -name '*_results.db' | head -1)" "$OUT/${TAG}_kernel_trace_stats_cartpole${suffix}.txt" > /dev/null
    rm -rf "$OUT/${TAG}_kt${suffix}"
done

# 4. SQ counters of the step kernel (instruction counts, busy / wait cycles), one pass
rm -rf "$OUT/${TAG}_pmc_sq"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    -d "$OUT/${TAG}_pmc_sq" -o r -- python "$REPO/bench.py" --steps 100 --warmup 20 --cpu-seconds 0 --no-probe --repetitions 2 > /dev/null 2> "$OUT/${TAG}_pmc_sq.err"
python "$REPO/tools/summarize_rocprof.py" counters "$(find $OUT/${TAG}_pmc_sq -name '*_results.db' | head -1)" "$OUT/${TAG}_pmc_sq.txt" step_kernel \
    "SQ PMC counters of the CartPole step kernel at 2^20 lanes (rocprofv3 --pmc, one pass), kernel sources $SHA" > /dev/null
rm -rf "$OUT/${TAG}_pmc_sq"
rm -rf "$OUT/${TAG}_pmc_sq2"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES \
    -d "$OUT/${TAG}_pmc_sq2" -o r -- python "$REPO/bench.py" --steps 100 --warmup 20 --cpu-seconds 0 --no-probe --repetitions 2 > /dev/null 2> "$OUT/${TAG}_pmc_sq2.err"
python "$REPO/tools/summarize_rocprof.py" counters "$(find $OUT/${TAG}_pmc_sq2 -name '*_results.db' | head -1)" "$OUT/${TAG}_pmc_sq_mem.txt" step_kernel \
    "SQ memory-instruction counters of the CartPole step kernel at 2^20 lanes, kernel sources $SHA" > /dev/null
rm -rf "$OUT/${TAG}_pmc_sq2"

# 5. fused rollout: bench lines and SQ_INSTS_VALU per launch (the numerator of its VALU-issue roofline)
for env in cartpole mountain_car pendulum; do
    n=$([ $env = pendulum ] && echo 4194304 || echo 1048576)
    for rec in "" "--record"; do
        key=${env}${rec:+_recorded}
        rm -rf "$OUT/${TAG}_pmc_valu_${key}"
        rocprofv3 --pmc SQ_INSTS_VALU -d "$OUT/${TAG}_pmc_valu_${key}" -o r -- python "$REPO/bench.py" --env $env --rollout 128 $rec --steps 256 --warmup 128 \
            --cpu-seconds 0 --repetitions 2 > /dev/null 2> "$OUT/${TAG}_pmc_valu_${key}.err"
    done
done
cd "$REPO"
python tools/summarize_rocprof.py valu "$(find $OUT/${TAG}_pmc_valu_cartpole -name '*_results.db' | head -1)" cartpole 1048576 128 $SHA profiles/pmc_valu.json > /dev/null
for env in mountain_car pendulum; do
    n=$([ $env = pendulum ] && echo 4194304 || echo 1048576)
    python tools/summarize_rocprof.py valu "$(find $OUT/${TAG}_pmc_valu_${env} -name '*_results.db' | head -1)" $env $n 128 $SHA profiles/pmc_valu.json > /dev/null
done
for env in cartpole mountain_car pendulum; do
    n=$([ $env = pendulum ] && echo 4194304 || echo 1048576)
    python tools/summarize_rocprof.py valu "$(find $OUT/${TAG}_pmc_valu_${env}_recorded -name '*_results.db' | head -1)" ${env}_recorded $n 128 $SHA profiles/pmc_valu.json > /dev/null
done
cp profiles/pmc_valu.json "$OUT/${TAG}_pmc_valu.json"
rm -rf $OUT/${TAG}_pmc_valu_*/
for env in cartpole mountain_car pendulum; do
    python bench.py --env $env --rollout 128 --steps 2048 --warmup 256 --cpu-seconds 0 > "$OUT/${TAG}_bench_rollout_${env}.json" 2>/dev/null
    python bench.py --env $env --rollout 128 --record --steps 1024 --warmup 128 --cpu-seconds 0 > "$OUT/${TAG}_bench_rollout_recorded_${env}.json" 2>/dev/null
done

# 6. batch-size sweep against the in-library copy probe (DRAM-resident regime), small batches with graph replay
$PIN python tools/size_sweep.py --sizes 14,16,18,20,21,22,23,24,25 > "$OUT/${TAG}_size_sweep.log" 2>&1
$PIN python tools/size_sweep.py --sizes 20,22,23,24 --nt 1 >> "$OUT/${TAG}_size_sweep.log" 2>&1
$PIN python tools/size_sweep.py --sizes 20,22,23,24 --nt 2 >> "$OUT/${TAG}_size_sweep.log" 2>&1
for n in 1024 16384 131072; do
    for g in "" "--graph"; do
        python bench.py --n-envs $n --steps 2000 --warmup 400 --cpu-seconds 0 --no-probe $g 2>/dev/null
    done
done > "$OUT/${TAG}_small_batch_graph.jsonl"

# 7. A/B against the round-1 library if it was shipped (_ab/libgymrs_r01.so): same box, same call
$PIN python tools/exp_split_streams.py > "$OUT/${TAG}_split_streams.log" 2>&1
# 8. GYMRS_TIME_LIMIT elision (CartPole, all three flags): us per step, launches that ran without the limit, refreshes
$PIN python tools/exp_limit_elision.py > "$OUT/${TAG}_time_limit_elision.log" 2>&1
$PIN python tools/exp_exact_limit.py >> "$OUT/${TAG}_time_limit_elision.log" 2>&1
if [ -f _ab/libgymrs_r01.so ]; then
    for i in 1 2; do
        $PIN python tools/step_timer.py --lib _ab/libgymrs_r01.so --flags 7 --reps 5
        $PIN python tools/step_timer.py --flags 7 --reps 5
    done >> "$OUT/${TAG}_time_limit_elision.log" 2>&1
fi
if [ -f _ab/libgymrs_r01.so ]; then
    for i in 1 2 3; do
        $PIN python tools/step_timer.py --lib _ab/libgymrs_r01.so --reps 5
        $PIN python tools/step_timer.py --reps 5
    done > "$OUT/${TAG}_ab_vs_round1.log" 2>&1
fi
# 9. what the pinning is worth: the driver's command alternately pinned and not
bash tools/exp_cpu_pinning.sh > "$OUT/${TAG}_cpu_pinning.log" 2>&1
du -sh "$OUT" | tail -1
echo evidence-done
