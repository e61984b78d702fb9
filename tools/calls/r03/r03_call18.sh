#!/bin/bash
set -u
OUT=gpurun_out/r03_c18
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
SHA=$(python -c "import bench; print(bench.kernel_source_sha16())")
cd /tmp
for set in "sq:SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "sq_mem:SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES"; do
    name=${set%%:*}; ctrs=${set#*:}
    rm -rf "$GRAFT_REPO_ROOT/$OUT/pmc_$name"
    GYMRS_AQL=0 rocprofv3 --pmc $ctrs -d "$GRAFT_REPO_ROOT/$OUT/pmc_$name" -o r -- python "$GRAFT_REPO_ROOT/bench.py" --steps 100 --warmup 20 --cpu-seconds 0 --no-probe --no-configs --repetitions 2 \
        > /dev/null 2> "$GRAFT_REPO_ROOT/$OUT/pmc_$name.err"
    python "$GRAFT_REPO_ROOT/tools/summarize_rocprof.py" counters "$(find $GRAFT_REPO_ROOT/$OUT/pmc_$name -name '*_results.db' | head -1)" "$GRAFT_REPO_ROOT/$OUT/r03_pmc_$name.txt" step_kernel \
        "PMC counters ($ctrs) of the CartPole step kernel at 2^20 lanes, HIP-launched (rocprofv3 --pmc cannot follow chains), kernel sources $SHA" > /dev/null
    rm -rf "$GRAFT_REPO_ROOT/$OUT/pmc_$name"
done
cd $GRAFT_REPO_ROOT
head -20 $OUT/r03_pmc_sq.txt
